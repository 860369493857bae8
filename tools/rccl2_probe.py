"""Development aid: what does RCCL say to two ranks on ONE device?  (tests/test_multigpu_gpu.py::test_rccl_two_ranks_... accepts either answer)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.multiprocessing as mp
from tests import test_multigpu_gpu as T
if __name__ == "__main__":
    ctx = mp.get_context("spawn"); ret = ctx.Queue(); port = T._free_port()
    procs = [ctx.Process(target=T._worker_rccl2, args=(rk, 2, port, ret)) for rk in range(2)]
    for p in procs: p.start()
    for _ in range(2): print(ret.get(timeout=180))
    for p in procs: p.join(timeout=30)
