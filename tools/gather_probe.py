import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29711", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
m = 48 << 20
src = torch.zeros(m, dtype=torch.uint8, device=dev)
recv = torch.empty((1, m), dtype=torch.uint8, device=dev)
host = torch.empty((1, m), dtype=torch.uint8, pin_memory=True)
print("pinned:", host.is_pinned())
st = torch.cuda.Stream(device=dev)
def T(label, t0):
    print(f"{label:28s} {1e3 * (time.perf_counter() - t0):9.3f} ms"); return time.perf_counter()
for it in range(3):
    print("--- iteration", it)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        t = time.perf_counter()
        mine = torch.tensor([m], dtype=torch.int64).to(dev, non_blocking=True); t = T("tensor.to", t)
        allsz = torch.empty(1, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allsz, mine); t = T("all_gather_into_tensor", t)
        sizes = allsz.tolist(); t = T("tolist", t)
        dist.gather(src[:m], [recv[0][:m]], dst=0); t = T("gather (issue)", t)
        st.synchronize(); t = T("gather (sync)", t)
        host[0][:m].copy_(recv[0][:m], non_blocking=True); t = T("D2H issue", t)
        st.synchronize(); t = T("D2H sync", t)
        # alternative gather: plain copy_ for own part
        recv[0][:m].copy_(src[:m], non_blocking=True); st.synchronize(); t = T("D2D copy_", t)
dist.destroy_process_group()
