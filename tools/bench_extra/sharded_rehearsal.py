import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from kreeq_amd import build
if len(sys.argv) > 1: build.LIB = os.path.abspath(sys.argv[1])
from kreeq_amd import synth
from kreeq_amd.dist import GpuEngine, ShardedCounter
g = synth.genome_codes(5_000_000, 1); r = synth.reads_batch(g, 1_000_000, 150, seed=2)
t = torch.from_numpy(r).cuda()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
eng = GpuEngine(21, 128, 0, capacity_hint=24_000_000); eng.db.set_option("trust_capacity", 1)
c = ShardedCounter(eng, 21, 128, sharded_path=True)
for i in range(3):
    eng.clear(); c.count_batch(t); eng.flush()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    eng.clear(); c.count_batch(t); eng.flush()        # flush: the table pass over the pending set belongs to the step (clear would drop it)
torch.cuda.synchronize()
s = c.summary()
print("sharded ms/step", (time.perf_counter() - t0) * 100, "ok" if (s["total"], s["distinct"]) == (130000000, 17733815) else ("WRONG", s))
