"""What can a Python thread observe about ProcessGroupNCCL's watchdog having dropped finished eager work?  (1-rank RCCL group)
   gpurun -- 'python tools/archive/r06/watchdog_probe.py'"""
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
if os.environ.get("PROBE_FR"):  # flight recorder on: its entries carry what the watchdog has seen
    os.environ["TORCH_NCCL_TRACE_BUFFER_SIZE"] = os.environ["PROBE_FR"]
    os.environ["TORCH_FR_BUFFER_SIZE"] = os.environ["PROBE_FR"]
import torch
import torch.distributed as dist

from obman_train_amd import dp

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dp.init_rccl(dev, rank=0, world_size=1)
pg = dist.distributed_c10d._get_default_group()
be = pg._get_backend(dev)
x = [torch.ones(1 << 20, device=dev) for _ in range(6)]
works = [dist.all_reduce(t, async_op=True) for t in x]
c = torch._C._distributed_c10d
print("has _dump_nccl_trace:", hasattr(c, "_dump_nccl_trace"), "json:", hasattr(c, "_dump_nccl_trace_json"))


def status():
    out = {}
    try:
        d = pickle.loads(c._dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
        out["keys"] = sorted(d.keys())
        out["pg_status"] = d.get("pg_status")
        ent = d.get("entries") or []
        out["entries"] = len(ent)
        if ent:
            out["states"] = [e.get("state") for e in ent][-8:]
            out["discovered"] = [e.get("time_discovered_completed_ns") is not None for e in ent][-8:]
            out["entry_keys"] = sorted(ent[-1].keys())
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)[:200]
    return out


print("t=0 before sync", status())
t0 = time.perf_counter()
torch.cuda.synchronize()
print("sync %.4f s" % (time.perf_counter() - t0), "is_completed:", [w.is_completed() for w in works])
print("seq:", be._get_sequence_number_for_group())
for i in range(30):
    s = status()
    print("t=%.3f" % (time.perf_counter() - t0), s.get("pg_status"), s.get("states"), s.get("discovered"), s.get("error"), s.get("entry_keys") if i == 0 else "")
    time.sleep(0.02)
dist.destroy_process_group()
