kreeq validate -f testFiles/random1.gfa -r testFiles/random1.fastq.gz testFiles/random2.fastq.gz
embedded
DBG Summary statistics:
Total kmers: 1572
Unique kmers: 13
Distinct kmers: 115
Missing kmers: 4398046510989
Total edges: 196
Missing	Total	QV	Error	k	Method
42	158	18.3545	0.0146068	21	Merqury
42	158	18.3545	0.0146068	21	Kreeq
