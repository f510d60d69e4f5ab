"""Worker of tests/test_gpu_rccl_preflight.py (its own process: a process group under real RCCL, world size 1, on cuda:0).

Drives detex_amd/sharding.py's two gathers -- incl. a sub-group -- and a grouped send / receive to itself (ncclSend / ncclRecv inside one
group call: the primitive gather_image_to_root's N > 1 branch is made of) with CUDA tensors under the `nccl` backend, every result
compared with the oracle.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import oracle_lib as ol
    from detex_amd import binding, formats as F, sharding

    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    binding.load()
    orc = ol.Oracle()
    out = {"backend": dist.get_backend(), "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "checks": {}}

    def decode(f, band, width, rows):
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        px = binding.decompress_linear_device(f, torch.from_numpy(np.ascontiguousarray(band)).cuda(), width, rows, status=status)
        torch.cuda.synchronize()
        return bool(status.item() == 0), px

    ones = torch.ones(1, dtype=torch.int32, device="cuda")
    dist.all_reduce(ones)
    out["checks"]["all_reduce"] = int(ones.item()) == 1
    sub = dist.new_group([0], backend="nccl")
    for name, (w, h) in (("BC1", (512, 256)), ("BPTC", (256, 260)), ("BPTC_FLOAT", (128, 68))):
        fmt = F.BY_NAME[name]
        wb, hb = (w + 3) // 4, (h + 3) // 4
        data = ol.stream_u(fmt, wb * hb, seed=0x2CC1 + fmt.index)
        ok_ref, want = orc.linear(fmt, data, w, h)
        for label, group in (("world", None), ("subgroup", sub)):
            shard, ok, local = sharding.decode_shard(decode, fmt, data, w, h, 0, 1)
            ok_all, image = sharding.gather_image(dist, torch, fmt, w, h, shard, local, ok, group=group)
            out["checks"]["%s/%s/gather_image" % (name, label)] = bool(ok_all == ok_ref and np.array_equal(image.cpu().numpy(), want))
            ok_root, image = sharding.gather_image_to_root(dist, torch, fmt, w, h, shard, local, ok, root=0, group=group)
            out["checks"]["%s/%s/gather_image_to_root" % (name, label)] = bool(ok_root == ok_ref and np.array_equal(image.cpu().numpy(), want))
    # grouped point-to-point under RCCL: two bands of an image sent to and received from this very rank inside ONE batch (ncclGroupStart ...
    # ncclSend / ncclRecv ... ncclGroupEnd), straight into slices of the destination image, as the root of an N-rank gather receives them
    fmt = F.BY_NAME["BC3"]
    w, h = 1024, 512
    data = ol.stream_u(fmt, (w // 4) * (h // 4), seed=0x5E1F)
    _, local = decode(fmt, data, w, h)
    image = torch.zeros_like(local)
    half = local.numel() // 2
    ops = []
    for lo in (0, half):
        ops.append(dist.P2POp(dist.irecv, image[lo:lo + half], 0))
        ops.append(dist.P2POp(dist.isend, local[lo:lo + half], 0))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    torch.cuda.synchronize()
    out["checks"]["grouped_send_recv_to_self"] = bool(torch.equal(image, local)) and bool(np.array_equal(image.cpu().numpy(), orc.linear(fmt, data, w, h)[1]))
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    out["ok"] = all(out["checks"].values())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
