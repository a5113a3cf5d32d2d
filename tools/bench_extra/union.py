import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from kreeq_amd import synth, KreeqDB
g = synth.genome_codes(5_000_000, 1)
dbs = []
for seed in (2, 3):
    r = synth.reads_batch(g, 1_000_000, 150, seed=seed)
    t = torch.from_numpy(r).cuda()
    db = KreeqDB(21, 128, capacity_hint=24_000_000); db.count_batch_dev(t.data_ptr(), t.numel()); db.sync(); dbs.append(db)
n = [d.summary()["distinct"] for d in dbs]
for path in ('direct', 'partitioned'):
  for rep in range(2):
    dst = KreeqDB(21, 128, capacity_hint=40_000_000); dst.set_option('merge_path', path)
    dst.sync(); t0 = time.perf_counter(); dst.merge(dbs[0]); dst.sync(); t1 = time.perf_counter(); dst.merge(dbs[1]); dst.sync(); t2 = time.perf_counter()
    print(path, f"merge into empty: {n[0]/(t1-t0)/1e9:.2f} G entries/s ({(t1-t0)*1e3:.2f} ms); merge into filled: {n[1]/(t2-t1)/1e9:.2f} G entries/s ({(t2-t1)*1e3:.2f} ms); union distinct {dst.summary()['distinct']}")
e = dbs[0].export()
t0 = time.perf_counter(); d2 = KreeqDB(21, 128, capacity_hint=24_000_000); d2.import_entries(e); d2.sync(); t1 = time.perf_counter()
print(f"import (host entries, PCIe incl.): {len(e)/(t1-t0)/1e9:.3f} G entries/s")
