kreeq validate -f testFiles/random2.fastq.gz -r testFiles/random1.fastq testFiles/random2.fastq
embedded
DBG Summary statistics:
Total kmers: 1572
Unique kmers: 13
Distinct kmers: 115
Missing kmers: 4398046510989
Total edges: 196
Missing	Total	QV	Error	k	Method
0	1400	inf	0	21	Merqury
0	1400	inf	0	21	Kreeq
