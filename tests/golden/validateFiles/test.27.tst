kreeq validate -f testFiles/random2.fastq.gz -r testFiles/random1.fastq.gz
embedded
DBG Summary statistics:
Total kmers: 172
Unique kmers: 25
Distinct kmers: 96
Missing kmers: 4398046511008
Total edges: 160
Missing	Total	QV	Error	k	Method
370	1400	18.3837	0.0145086	21	Merqury
370	1400	18.3837	0.0145086	21	Kreeq
