kreeq validate -f testFiles/random3.N.fastq -r testFiles/random1.fastq
embedded
DBG Summary statistics:
Total kmers: 172
Unique kmers: 25
Distinct kmers: 96
Missing kmers: 4398046511008
Total edges: 160
Missing	Total	QV	Error	k	Method
0	27	inf	0	21	Merqury
0	27	inf	0	21	Kreeq
