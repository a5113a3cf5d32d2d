kreeq validate -f testFiles/random1.fastq.gz -r testFiles/random2.fastq
embedded
DBG Summary statistics:
Total kmers: 1400
Unique kmers: 0
Distinct kmers: 79
Missing kmers: 4398046511025
Total edges: 138
Missing	Total	QV	Error	k	Method
61	172	16.853	0.0206395	21	Merqury
61	172	16.853	0.0206395	21	Kreeq
