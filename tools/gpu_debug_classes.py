#!/usr/bin/env python3
"""GPU debugging aid: per-class mismatch report of the device decoders vs the CPU oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from detex_amd import binding, formats as F
import oracle_lib as ol, streams
orc = ol.Oracle()
names = sys.argv[1:] or [f.name for f in F.FORMATS]
for name in names:
    f = F.BY_NAME[name]
    for label, blocks in streams.forced_classes(f):
        n = len(blocks)
        out, ok = binding.decompress_blocks_device(f, torch.from_numpy(blocks.reshape(-1)).cuda(), n)
        torch.cuda.synchronize()
        out = out.cpu().numpy().reshape(n, -1); ok = ok.cpu().numpy()[:n].astype(bool)
        ok_o, out_o = orc.blocks(f, blocks)
        bad = np.nonzero((ok != ok_o) | (out != out_o).any(axis=1))[0]
        if len(bad):
            i = bad[0]
            print("%s/%s: %d/%d bad; first %d ok gpu/orc %s/%s in %s" % (name, label, len(bad), n, i, ok[i], ok_o[i], blocks[i].tobytes().hex()))
            print("   gpu", out[i][:32].tolist()); print("   orc", out_o[i][:32].tolist())
    print(name, "done", flush=True)
