"""Host-side profile (cProfile) of the train loop fed from host memory: is the loop host-bound?
python tools/feed_hostprof.py [wire|compact] [steps]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch  # noqa: E402

import bench  # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else 'wire'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sys.argv = sys.argv[:1]
a = bench.parse()
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
bench.train_loop_rates(a, dev, steps=40, warm=10, legs=(leg,))     # everything compiled / recorded
pr = cProfile.Profile()
pr.enable()
out = bench.train_loop_rates(a, dev, steps=steps, warm=steps // 4, legs=(leg,))
pr.disable()
print(out)
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
