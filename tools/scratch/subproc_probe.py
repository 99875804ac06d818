import subprocess, sys, torch
print("parent cuda:", torch.cuda.is_available(), torch.zeros(1, device="cuda").item())
r = subprocess.run([sys.executable, "-c", "import torch; print('child cuda:', torch.cuda.is_available(), torch.ones(1, device='cuda').item())"], capture_output=True, text=True)
print("child rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
