"""Block-row sharding of one texture across ranks.

The decode itself needs no exchange: blocks are stored row-major (texture.c:115-141) and the
linear image is row-major with pitch width*px (texture.c:129-131), so the shard of rank g --
block rows [g*hb/G, (g+1)*hb/G) -- is ONE contiguous byte range of the input and ONE contiguous
byte range of the output.  The only communication on this path is the optional whole-image gather: to one rank (gather_image_to_root: grouped
point-to-point sends straight into the root's image) or to every rank (gather_image: one all_gather_into_tensor);
decode throughput never includes it.
"""
from collections import namedtuple

Shard = namedtuple("Shard", "rank world row0 row1 in_offset in_bytes out_offset out_bytes px_row0 px_rows")


def shard_rows(rank, world, height_in_blocks):
    """block rows [row0, row1) owned by `rank`: balanced to within one row, in order."""
    return rank * height_in_blocks // world, (rank + 1) * height_in_blocks // world


def shard_of(rank, world, fmt, width, height, pitch=None):
    """Byte ranges of rank's band of a width x height texture of format fmt."""
    wb, hb = (width + 3) // 4, (height + 3) // 4
    px = fmt.pixel_bytes
    pitch = width * px if pitch is None else pitch
    r0, r1 = shard_rows(rank, world, hb)
    y0, y1 = min(r0 * 4, height), min(r1 * 4, height)
    return Shard(rank, world, r0, r1, r0 * wb * fmt.block_bytes, (r1 - r0) * wb * fmt.block_bytes,
                 y0 * pitch, (y1 - y0) * pitch, y0, y1 - y0)


def decode_shard(decode_fn, fmt, blocks, width, height, rank, world):
    """Decode this rank's band.  decode_fn(fmt, band_blocks, width, band_height) -> (ok, pixels)
    is the device-tier call on the GPU box (binding.decompress_linear_device) or the oracle in
    the CPU tests.  `blocks` is the whole texture's block stream (any buffer sliceable by byte)."""
    s = shard_of(rank, world, fmt, width, height)
    band = blocks[s.in_offset:s.in_offset + s.in_bytes]
    ok, pixels = decode_fn(fmt, band, width, s.px_rows)
    return s, ok, pixels


def gather_layout(world, fmt, width, height):
    """Equal-size chunk layout of the optional whole-image gather: rank r's band starts at
    r*chunk in a world*chunk staging image, chunk = the largest band.  Bands differ by at most
    one block row, so when height_in_blocks % world == 0 (every BASELINE config) the staging
    image IS the final image and the gather is a single collective with no copy before or after."""
    sizes = [shard_of(r, world, fmt, width, height).out_bytes for r in range(world)]
    chunk = max(sizes) if sizes else 0
    exact = all(shard_of(r, world, fmt, width, height).out_offset == r * chunk for r in range(world)) and \
        sum(sizes) == world * chunk
    return chunk, sizes, exact


def gather_image(dist, torch, fmt, width, height, shard, local_pixels, local_ok=True, group=None, image=None):
    """Optional whole-image gather: every rank ends up with the full row-major image and the AND
    of the per-rank ok flags (the reference's bool result, texture.c:144).

    ONE all_gather_into_tensor (RCCL on the GPU box: direct peer transfers over xGMI; gloo in the
    CPU tests) of equal chunks, received straight into the final image when the bands are equal
    (gather_layout(...).exact); otherwise the short bands are padded at the tail and the image is
    compacted once.  The ok flags travel as one extra all_reduce(MIN) of a single byte-sized
    tensor.  `image` may be a preallocated world*chunk (exact: width*height*px) uint8 tensor."""
    world = dist.get_world_size(group)
    px = fmt.pixel_bytes
    chunk, sizes, exact = gather_layout(world, fmt, width, height)
    dev = local_pixels.device
    flat = local_pixels.reshape(-1)
    if flat.numel() == chunk:
        send = flat
    else:                                   # a band one block row short: pad the tail
        send = torch.zeros(chunk, dtype=torch.uint8, device=dev)
        send[:shard.out_bytes] = flat[:shard.out_bytes]
    if image is None or image.numel() != world * chunk:
        image = torch.empty(world * chunk, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(image, send, group=group)
    flag = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if not exact:
        out = torch.empty(width * height * px, dtype=torch.uint8, device=dev)
        for r in range(world):
            s = shard_of(r, world, fmt, width, height)
            out[s.out_offset:s.out_offset + s.out_bytes] = image[r * chunk:r * chunk + s.out_bytes]
        image = out
    return bool(flag.item()), image


def gather_image_to_root(dist, torch, fmt, width, height, shard, local_pixels, local_ok=True, root=0, group=None, image=None):
    """Optional whole-image gather to ONE rank: every other rank sends its band straight into its place in
    the root's image -- grouped point-to-point transfers (ncclSend / ncclRecv inside one group on the GPU box, so the
    root's xGMI links to all peers carry data at once; gloo in the CPU tests), no staging copy, bands of any (unequal)
    size.  Only 1/world of what an all-gather moves crosses each link, and nothing lands on the other ranks.
    Returns (ok, image) on the root -- ok = AND of the per-rank flags, the reference's bool result (texture.c:144) --
    and (ok, None) elsewhere.  `image` may be a preallocated width*height*px uint8 tensor on the root.  `root` and the shard
    ranks are ranks within `group` (the default group when None)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # ranks below are ranks IN THE GROUP (as `root` is); P2POp addresses its peer by GLOBAL rank
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    px = fmt.pixel_bytes
    dev = local_pixels.device
    flat = local_pixels.reshape(-1)[:shard.out_bytes]
    ops = []
    if rank == root:
        if image is None or image.numel() != width * height * px:
            image = torch.empty(width * height * px, dtype=torch.uint8, device=dev)
        for r in range(world):
            s = shard_of(r, world, fmt, width, height)
            if s.out_bytes == 0:
                continue
            if r == root:
                image[s.out_offset:s.out_offset + s.out_bytes] = flat
            else:
                ops.append(dist.P2POp(dist.irecv, image[s.out_offset:s.out_offset + s.out_bytes], peer(r), group))
    elif shard.out_bytes:
        ops.append(dist.P2POp(dist.isend, flat.contiguous(), peer(root), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    flag = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item()), (image if rank == root else None)
