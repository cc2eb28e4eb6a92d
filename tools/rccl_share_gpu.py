"""Probe: does RCCL accept two ranks on ONE GPU?  (If it does, the executor's
exchange can be tested with a real 2-rank communicator on a 1-GPU box.)
Parent spawns 2 children; each sets device 0, joins an nccl group of 2 and
all-reduces rank+1."""
import os
import subprocess
import sys

if 'RANK' not in os.environ:
    kids = [subprocess.Popen([sys.executable, __file__], env=dict(
        os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
        MASTER_PORT='29731', HSA_ENABLE_IPC_MODE_LEGACY='0')) for r in range(2)]
    rcs = []
    for k in kids:
        try:
            rcs.append(k.wait(timeout=150))
        except subprocess.TimeoutExpired:
            k.kill()
            rcs.append('timeout')
    print('rcs', rcs)
    sys.exit(0)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
import datetime
dist.init_process_group('nccl', rank=int(os.environ['RANK']), world_size=2,
                        timeout=datetime.timedelta(seconds=90), device_id=torch.device('cuda', 0))
t = torch.full((1024,), float(int(os.environ['RANK']) + 1), device='cuda')
dist.all_reduce(t)
torch.cuda.synchronize()
print('rank', os.environ['RANK'], 'sum', float(t[0]), flush=True)
dist.destroy_process_group()
