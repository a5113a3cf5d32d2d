import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from kreeq_amd import synth, KreeqDB
g = synth.genome_codes(5_000_000, 1); r = synth.reads_batch(g, 1_000_000, 150, seed=2)
t = torch.from_numpy(r).cuda()
ga = torch.from_numpy(synth.codes_to_ascii(synth.mutate(g, 1e-4, seed=3))).cuda()
db = KreeqDB(21, 128, capacity_hint=24_000_000); db.set_option("trust_capacity", 1)
s = torch.cuda.Stream(); db.set_stream(s.cuda_stream)
db.count_batch_dev(t.data_ptr(), t.numel()); db.sync()
ctr = torch.zeros(3, dtype=torch.int64, device="cuda")
pb = torch.zeros(t.numel() * 16, dtype=torch.uint8, device="cuda")
for path in ("direct", "partitioned"):
  db.set_option("lookup_path", path)
  for name, seq, per_base in (("reads-as-assembly", t, False), ("genome(1e-4 subst)", ga, False)) + ((("reads-as-assembly per-base", t, True),) if path == "direct" else ()):
      for i in range(2):
          db.lookup_sequence_dev(seq.data_ptr(), seq.numel(), ctr.data_ptr(), per_base_ptr=pb.data_ptr() if per_base else None)
      torch.cuda.synchronize(); ctr.zero_(); torch.cuda.synchronize()
      t0 = time.perf_counter(); n = 5
      for i in range(n):
          db.lookup_sequence_dev(seq.data_ptr(), seq.numel(), ctr.data_ptr(), per_base_ptr=pb.data_ptr() if per_base else None)
      torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
      c = (ctr // n).cpu().tolist()
      print(f"{path} {name}: {dt*1e3:.3f} ms  {c[1]/dt/1e9:.1f} G k-mers/s  algorithmic {c[1]*(34 if per_base else 18)/dt/1e9:.0f} GB/s  counters {c}")
