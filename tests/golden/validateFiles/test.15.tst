kreeq validate -f testFiles/random1.gfa -r testFiles/random1.fastq
embedded
DBG Summary statistics:
Total kmers: 172
Unique kmers: 25
Distinct kmers: 96
Missing kmers: 4398046511008
Total edges: 160
Missing	Total	QV	Error	k	Method
63	158	16.2099	0.0239336	21	Merqury
63	158	16.2099	0.0239336	21	Kreeq
