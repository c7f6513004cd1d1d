import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, zlib
from crumble_amd import gz
import test_gz as T
c = gz.GzCodec(0)
rng = np.random.default_rng(11)
for n in (1, 7, 700, 70000):
    for kind, data in T.samples(rng, n).items():
        for level in (0, 1, 6):
            comp = T.bgzf(data, level)
            for rep in range(2):
                got = c.inflate(comp).tobytes()
                ok = got == data
                if not ok:
                    diff = [i for i in range(len(data)) if got[i] != data[i]]
                    print("BAD", kind, n, level, rep, len(diff), diff[:8], got[:8].hex(), data[:8].hex())
                    break
print("done")
c.close()
