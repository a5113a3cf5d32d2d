kreeq validate -f testFiles/random1.fasta -r testFiles/random1.fastq.gz
embedded
DBG Summary statistics:
Total kmers: 172
Unique kmers: 25
Distinct kmers: 96
Missing kmers: 4398046511008
Total edges: 160
Missing	Total	QV	Error	k	Method
177	711	18.6839	0.0135397	21	Merqury
177	711	18.6839	0.0135397	21	Kreeq
