"""measurement: bench.py's env_api legs (the drop-in loop `while not env.step()`) printed as one line: step / resident ms per step"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else 'c2,c3'
out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--env-api-only', which], capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
e = d.get('env_api', d)
print(' '.join(f"{k}:{v['step']['ms_per_step']:.4f}/{v['resident_ms_per_step']:.4f}" for k, v in e.items() if isinstance(v, dict)))
