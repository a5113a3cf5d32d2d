kreeq validate -f testFiles/random1.fastq -r testFiles/random1.fastq.gz testFiles/random2.fastq.gz
embedded
DBG Summary statistics:
Total kmers: 1572
Unique kmers: 13
Distinct kmers: 115
Missing kmers: 4398046510989
Total edges: 196
Missing	Total	QV	Error	k	Method
0	172	inf	0	21	Merqury
0	172	inf	0	21	Kreeq
