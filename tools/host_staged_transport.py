"""A transport for recommend_amd.dist.set_transport that lets several ranks share ONE GPU: RCCL
refuses two ranks on one device, so rehearsals and the two-rank GPU tests use a gloo process
group, and gloo moves host memory -- every collective here copies the device tensors to the
host, runs the collective there and copies the result back.  Test scaffolding: production (one
rank per GPU, backend "nccl") keeps the default transport and never imports this file."""
import torch
import torch.distributed as dist


class HostStagedTransport:
    @staticmethod
    def collective_device(device):
        return torch.device("cpu")

    @staticmethod
    def all_gather(out_list, t, group):
        c = t.cpu()
        got = [torch.empty_like(c) for _ in out_list]
        dist.all_gather(got, c, group=group)
        for o, g in zip(out_list, got):
            o.copy_(g)

    @staticmethod
    def all_gather_into(out, t, group):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, t.cpu().contiguous(), group=group)
        out.copy_(host)

    @staticmethod
    def reduce_scatter(out, full, group):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(host, full.cpu().contiguous(), group=group)
        out.copy_(host)

    @staticmethod
    def all_reduce(t, op, group):
        c = t.cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)

    @staticmethod
    def all_to_all(out, src, out_splits, in_splits, group):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, src.cpu().contiguous(), out_splits, in_splits, group=group)
        out.copy_(host)
