"""Golden-vector generator for the 3-D Dubins planner (runs only in the build container; imports /root/reference).
Writes tests/golden/F7_dubins_kat.json: the paper instance of dubinsmaneuver3d.py:230, the __main__ instance, 40 random."""
import sys
sys.path.insert(0,'/root/reference')
import matplotlib; matplotlib.use('Agg')
import numpy as np, math, json
from mamp.policies.sca import dubinsmaneuver3d as d3
rng=np.random.default_rng(5)
cases=[]
# paper instance and __main__ instance (dubinsmaneuver3d.py:194-248)
cases.append(([-80.0, 10.0, 250.0, float(np.deg2rad(20.0)), float(np.deg2rad(0.0))],[50.0, 70.0, 0.0, float(np.deg2rad(240.0)), float(np.deg2rad(0.0))],40,[float(np.deg2rad(-15.0)), float(np.deg2rad(20.0))]))
cases.append(([0.0, 0.0, 3.0, float(np.deg2rad(-90)), float(np.deg2rad(0.0))],[0.0, 0.0, 13.0, float(np.deg2rad(90)), float(np.deg2rad(0.0))],1.5,[float(np.deg2rad(-45.0)), float(np.deg2rad(45.0))]))
for _ in range(40):
    qi=list(rng.uniform(-20,20,3))+[rng.uniform(0,2*math.pi), rng.uniform(-0.5,0.5)]
    qf=list(rng.uniform(-20,20,3))+[rng.uniform(0,2*math.pi), rng.uniform(-0.5,0.5)]
    cases.append((qi,qf,1.5,[-math.pi/4,math.pi/4]))
out=[]
for qi,qf,R,pl in cases:
    m=d3.dubinsmaneuver3d(np.array(qi+[0.0]),np.array(qf+[0.0]),R,pl)
    out.append(dict(qi=qi,qf=qf,R=R,pl=pl,length=m.length,mode=m.mode,n=len(m.path),first=m.path[0],mid=m.path[len(m.path)//2],last=m.path[-1]))
json.dump(out,open('/root/repo/tests/golden/F7_dubins_kat.json','w'))
print(out[0]['mode'],out[0]['length'],out[0]['n'], np.deg2rad(100)==100.0*(math.pi/180.0))
