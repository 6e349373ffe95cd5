"""The process-group calls of bench.py's N > 1 entry, run with ONE rank on one GPU: default gloo group, PCI identity
all-gather, an RCCL sub-group created with device_id, the probe all-reduce, barrier(group, device_ids) and the MAX
all-reduce on a device tensor.  Checks the API usage of the installed torch where no second GPU exists."""
import datetime
import os

os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29621")
import torch
import torch.distributed as dist

dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
pr = torch.cuda.get_device_properties(0)
ident = (f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', -1):02x}:{getattr(pr, 'pci_device_id', -1):02x}", str(getattr(pr, "uuid", "")))
out = [None]
dist.all_gather_object(out, ident)
print("identity", out)
torch.cuda.set_device(0)
g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=60), device_id=torch.device("cuda", 0))
t = torch.ones(1, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.SUM, group=g)
torch.cuda.synchronize()
dist.barrier(group=g, device_ids=[0])
m = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(m, op=dist.ReduceOp.MAX, group=g)
print("rccl group ok", float(t.item()), float(m.item()))
dist.barrier()
dist.destroy_process_group()
