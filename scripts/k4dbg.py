import re,sys,subprocess,os
env=dict(os.environ, TEASER_K4_DEBUG="1")
out=subprocess.run([sys.executable,"scripts/profile_config5.py","batch"],env=env,capture_output=True,text=True).stderr
lines=[l for l in out.splitlines() if "exact search: problem" in l]
last=lines[-63:]
nodes=[int(re.search(r"nodes (\d+)",l).group(1)) for l in last]
n2=[int(re.search(r"n2 (\d+)",l).group(1)) for l in last]
ms=[float(re.search(r"problems: ([\d.]+) ms",l).group(1)) for l in last]
print("problems",len(last),"nodes total",sum(nodes),"max",max(nodes),"sorted top",sorted(nodes)[-8:],"n2 max",max(n2),"launch ms",ms[0])
print([l for l in out.splitlines() if "persistent waves" in l][-1][:300])
