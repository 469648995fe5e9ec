"""The flag-ordered tail against the stream-ordered schedule, bit for bit, over a sweep of sizes (round 6: after the change of
the task lists).  Two child processes per size (tools/fit_digest.py), alpha / log det / mean / sigma compared with array_equal.
usage: python tools/flow_bits_sweep.py [N ...]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(a) for a in sys.argv[1:]] or [5120, 5250, 5633, 6000, 6500, 7000, 7681, 8192, 9001, 10240, 12288, 14000]
tool = os.path.join(ROOT, "tools", "fit_digest.py")
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for n in sizes:
        outs = []
        for k, extra in enumerate(({"GPMI_FLOW": "0"}, {})):
            out = os.path.join(tmp, f"d{k}.npz")
            r = subprocess.run([sys.executable, tool, out, str(n)], env=dict(os.environ, **extra), capture_output=True, text=True)
            if r.returncode != 0:
                print(n, extra, "FAILED", r.stderr[-500:])
                bad += 1
                break
            outs.append(dict(np.load(out)))
        else:
            same = all(np.array_equal(outs[0][q], outs[1][q]) for q in outs[0])
            print(f"N={n}: {'bit-identical' if same else 'DIFFERENT'}")
            bad += not same
sys.exit(1 if bad else 0)
