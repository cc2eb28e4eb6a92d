"""Train loop fed from host memory (bench.train_loop_rates, one leg) for a trace:
python tools/feed_trace.py [wire|compact] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch  # noqa: E402

import bench  # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else 'wire'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sys.argv = sys.argv[:1]
a = bench.parse()
torch.cuda.set_device(0)
print(json.dumps(bench.train_loop_rates(a, torch.device('cuda:0'), steps=steps, warm=steps // 3,
                                        legs=(leg,))))
