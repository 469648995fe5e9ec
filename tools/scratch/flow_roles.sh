#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" python3 tools/config_bench.py cfg2 2>&1 | tail -1
  env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   headline ms_per_step', round(d['ms_per_step'],3), 'flow tail', round(d['roofline']['flow_tail']['ms_per_step'],3))"
}
run GPMI_FLOW_WINDOW=16 GPMI_FLOW_NEAR_D=8 GPMI_FLOW_NEAR_WGS=64
run GPMI_FLOW_WINDOW=16 GPMI_FLOW_NEAR_D=12 GPMI_FLOW_NEAR_WGS=96
run GPMI_FLOW_WINDOW=16 GPMI_FLOW_NEAR_D=16 GPMI_FLOW_NEAR_WGS=96
run GPMI_FLOW_WINDOW=4 GPMI_FLOW_NEAR_D=8 GPMI_FLOW_NEAR_WGS=64
python3 - <<PY
import numpy as np,struct
raw=open("gpurun_out/flow8k/trace.bin","rb").read()
m,nl,ntasks,cw=struct.unpack("4q",raw[:32]); p=32
off=np.frombuffer(raw,np.int32,nl+1,p); p+=4*(nl+1)
tasks=np.frombuffer(raw,np.dtype([("type","u1"),("s","u1"),("fadd","u1"),("pad","u1"),("i","u2"),("j","u2"),("k","u2"),("pad2","u2")]),ntasks,p); p+=12*ntasks
tr=np.frombuffer(raw,np.uint64,4*ntasks+m*cw,p).astype(np.int64)
tt=tr[:4*ntasks].reshape(ntasks,4)
Z=tasks["type"]==2
d=(tt[Z,2]-tt[Z,1])*0.01
print("Z body histogram (us):", np.histogram(d,bins=[0,50,60,70,80,90,100,110,120,130,150,200,400])[0])
PY
