kreeq union -d testFiles/test1.kreeq testFiles/test2.kreeq 
embedded
DBG Summary statistics:
Total kmers: 1572
Unique kmers: 13
Distinct kmers: 115
Missing kmers: 4398046510989
Total edges: 196
