kreeq validate -f testFiles/random1.gfa -r testFiles/random2.fastq
embedded
DBG Summary statistics:
Total kmers: 1400
Unique kmers: 0
Distinct kmers: 79
Missing kmers: 4398046511025
Total edges: 138
Missing	Total	QV	Error	k	Method
42	158	18.3545	0.0146068	21	Merqury
42	158	18.3545	0.0146068	21	Kreeq
