"""Block-row sharding of one texture across ranks.

The decode itself needs no exchange: blocks are stored row-major (texture.c:115-141) and the
linear image is row-major with pitch width*px (texture.c:129-131), so the shard of rank g --
block rows [g*hb/G, (g+1)*hb/G) -- is ONE contiguous byte range of the input and ONE contiguous
byte range of the output.  The only collective on this path is the optional whole-image gather
(all_gather of variable-size row bands, done as equal-size padded chunks or per-rank broadcasts),
used by callers that want the full image on every rank; decode throughput never includes it.
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


def gather_image(dist, torch, fmt, width, height, shard, local_pixels, local_ok=True, group=None):
    """Optional whole-image gather: every rank ends up with the full row-major image and the AND
    of the per-rank ok flags (the reference's bool result, texture.c:144).  Bands differ by at
    most one block row, so the exchange is one all_gather of equal-size padded chunks (RCCL on the
    GPU box: direct peer sends over xGMI; gloo in the CPU tests)."""
    world = dist.get_world_size(group)
    px = fmt.pixel_bytes
    sizes = [shard_of(r, world, fmt, width, height).out_bytes for r in range(world)]
    chunk = max(sizes) if sizes else 0
    send = torch.zeros(chunk + 1, dtype=torch.uint8, device=local_pixels.device)
    send[:shard.out_bytes] = local_pixels.reshape(-1)[:shard.out_bytes]
    send[chunk] = 1 if local_ok else 0
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    image = torch.empty(width * height * px, dtype=torch.uint8, device=local_pixels.device)
    ok = True
    for r in range(world):
        s = shard_of(r, world, fmt, width, height)
        image[s.out_offset:s.out_offset + s.out_bytes] = recv[r][:s.out_bytes]
        ok = ok and bool(recv[r][chunk].item())
    return ok, image
