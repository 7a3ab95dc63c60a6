import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zk_cryptography_amd as zk
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda n: torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
cm = zk.ComposedMultilinear([rnd(n) for _ in range(2)])
sc = zk.ComposedSumcheck(cm)
for _ in range(6): sc.prove()
torch.cuda.synchronize()
