"""Teardown probe (debugging aid): fit at size n, then leave the interpreter in different ways."""
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(25, exit=True)
sys.path[:0]=["/root/repo","/root/repo/inference-tools_amd"]
import numpy as np, workloads as wl
from inference_amd import _lib
from inference_amd.gp import GpRegressor
n, mode = int(sys.argv[1]), sys.argv[2]
if mode == "noatexit":
    import atexit
    _lib.load(); atexit.unregister(_lib._close_all_handles)
x,y,e = wl.synthetic_dataset(65, n, 5)
gp = GpRegressor(x,y,y_err=e,hyperpars=wl.timing_theta(wl.SE,y,5))
print("fit done", mode, flush=True)
if mode == "sleepclose":
    time.sleep(0.5); t=time.time(); gp.engine.close(); print("close after sleep", time.time()-t, flush=True)
if mode == "syncclose":
    gp.engine.sync(); t=time.time(); gp.engine.close(); print("close after sync", time.time()-t, flush=True)
if mode == "close":
    t=time.time(); gp.engine.close(); print("close", time.time()-t, flush=True)
if mode == "raise":
    raise SystemExit("leaving with a live regressor")
print("falling off the end", flush=True)
