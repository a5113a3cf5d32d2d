kreeq validate -f testFiles/random1.fasta -r testFiles/random2.fastq
embedded
DBG Summary statistics:
Total kmers: 1400
Unique kmers: 0
Distinct kmers: 79
Missing kmers: 4398046511025
Total edges: 138
Missing	Total	QV	Error	k	Method
274	711	16.3995	0.0229115	21	Merqury
274	711	16.3995	0.0229115	21	Kreeq
