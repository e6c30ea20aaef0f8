import os, sys, random, datetime
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
from tests.helpers import build_product_cyclegan, load_golden_steps, golden_inputs
from tests.test_step_graph_gpu import _run
c = dict(load_golden_steps()["c64_default"]["config"]); c["pool_size"] = 3
os.environ["LOCAL_RANK"] = "0"
single = build_product_cyclegan(c)
want = _run(single, c, 5)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29588", rank=0, world_size=1, timeout=datetime.timedelta(minutes=2))
os.environ["GS_FORCE_DDP"] = "1"; os.environ["GS_DDP_GRAPH_COLLECTIVES"] = "1"
ddp = build_product_cyclegan(c)
got = _run(ddp, c, 5)
print("graph", ddp._graph is not None, "update graph", ddp._graph_update is not None, "collectives captured", ddp._graph_collectives)
ok = all(torch.equal(got[s][2], want[s][2]) and got[s][0] == want[s][0] for s in range(5))
print("bitwise equal to single process:", ok)
for s in range(5):
    print(s, max(abs(got[s][0][k] - want[s][0][k]) for k in want[s][0]))
dist.destroy_process_group()
