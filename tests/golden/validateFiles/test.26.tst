kreeq validate -f testFiles/random2.fastq.gz -r testFiles/random2.fastq
embedded
DBG Summary statistics:
Total kmers: 1400
Unique kmers: 0
Distinct kmers: 79
Missing kmers: 4398046511025
Total edges: 138
Missing	Total	QV	Error	k	Method
0	1400	inf	0	21	Merqury
0	1400	inf	0	21	Kreeq
