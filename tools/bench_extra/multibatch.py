"""Count several different read batches into ONE table (no clear in between): the first batch finds the table empty
(k_count_regions skips the image read), the later ones stream it in and out."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from kreeq_amd import synth, KreeqDB
g = synth.genome_codes(5_000_000, 1)
batches = [torch.from_numpy(synth.reads_batch(g, 1_000_000, 150, seed=s)).cuda() for s in (2, 3, 4, 5)]
db = KreeqDB(21, 128, capacity_hint=70_000_000); db.set_option("trust_capacity", 1)
st = torch.cuda.Stream(); db.set_stream(st.cuda_stream)
for rep in range(2):
    db.clear(); db.sync()
    ts = []
    for t in batches:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        db.count_batch_dev(t.data_ptr(), t.numel()); db.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    s = db.summary()
    print("per-batch ms", [round(x, 3) for x in ts], "total", s["total"], "distinct", s["distinct"])
