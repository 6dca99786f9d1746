import sys, runpy
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
sys.argv = ["host_loop_probe.py"] + sys.argv[1:]
runpy.run_path(__file__.replace("host_loop_probe_torch.py", "host_loop_probe.py"), run_name="__main__")
