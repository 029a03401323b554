cd /root/repo
export TMPDIR=/tmp
python - <<'PY'
import os, subprocess, sys, tempfile
ROOT='/root/repo'; sys.path.insert(0, ROOT)
import numpy as np
from laser_slam_amd import synth
n_az=16384; n=6; d='/tmp/trk'; os.makedirs(d, exist_ok=True)
exe=d+'/track_driver'
subprocess.check_call(["g++","-std=c++17","-O2","-I",ROOT+"/include","-I",ROOT+"/laser_slam_amd/cpp/include",ROOT+"/tests/cpp/track_driver.cpp","-o",exe,"-L",ROOT+"/laser_slam_amd","-llsgpu_icp","-Wl,-rpath,"+ROOT+"/laser_slam_amd"])
scene=synth.Scene(1234)
with open(d+"/poses.txt","w") as f:
    for i in range(n):
        T=synth.se3(0.8*i,0.05*i,synth.SENSOR_HEIGHT,yaw=np.deg2rad(2.0*i))
        synth.hdl64_scan(scene,T,n_az,10+i).tofile(d+f"/scan{i}.bin")
        R=(T@synth.se3(0.1,-0.05,0,yaw=np.deg2rad(0.5)))
        qw=np.sqrt(1+R[0,0]+R[1,1]+R[2,2])/2
        q=[qw,(R[2,1]-R[1,2])/(4*qw),(R[0,2]-R[2,0])/(4*qw),(R[1,0]-R[0,1])/(4*qw)]
        f.write("%d %s\n"%(100000000*i," ".join(repr(float(v)) for v in [*q,*R[:3,3]])))
PY
rm -rf gpurun_out/prof_trk
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /root/repo/gpurun_out/prof_trk -- /tmp/trk/track_driver /tmp/trk 6 /root/repo/tests/golden/icp_chain.yaml 3 16 > /root/repo/gpurun_out/trk.out 2> /root/repo/gpurun_out/trk.err)
grep icp_iterations gpurun_out/trk.out
db=$(find gpurun_out/prof_trk -name "*results.db" | head -1)
python profiles/summarize_rocpd.py $db | head -30
