"""Parity of the rejected A/B kernels (profiles/AB_RECORD.md).  They are NOT part of the product library: build
them with `make lib-ab` and run
    DETEXHIP_LIB=build/explib/libdetexhip_ab.so python -m pytest tests/test_ab_variants.py -m gpu
Without DETEXHIP_LIB pointing at an A/B build this module is skipped."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from detex_amd import formats as F

pytestmark = [pytest.mark.gpu, pytest.mark.skipif("_ab" not in os.path.basename(os.environ.get("DETEXHIP_LIB", "")),
                                                  reason="needs DETEXHIP_LIB=<A/B build of libdetexhip>")]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).cuda()


def _first_diff(got, want, unit):
    bad = np.flatnonzero(got != want)
    return "first mismatch at byte %d (unit %d)" % (bad[0], bad[0] // unit) if bad.size else "equal"


def test_bc1_tile4x4_variant_matches(torch_cuda, oracle):
    """the north_star tile-shape variant (A/B only) decodes identically"""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME["BC1"]
    W, H = 2048, 256
    data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=77)
    _, want = oracle.linear(fmt, data, W, H)
    binding.set_kernel_variant(1)
    try:
        out = binding.decompress_linear_device(fmt, _dev(torch, data), W, H)
        torch.cuda.synchronize()
    finally:
        binding.set_kernel_variant(0)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("name,variant", [("BPTC", 3), ("BPTC", 4), ("BPTC", 5), ("BPTC_FLOAT", 3), ("BPTC_SIGNED_FLOAT", 3), ("BC1", 2), ("BPTC_FLOAT", 2)])
def test_alternative_decoder_variants_match(name, variant, torch_cuda, oracle, forced_vectors):
    """the A/B decoder implementations (profiles/AB_RECORD.md) decode identically, forced classes included"""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    W, H = 2048, 512
    n = (W // 4) * (H // 4)
    forced = forced_vectors[name + "/in"].reshape(-1)
    data = np.concatenate([forced, ol.stream_u(fmt, n, seed=0xAB + variant)])[:n * fmt.block_bytes]
    ok_o, want = oracle.linear(fmt, data, W, H)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    binding.set_kernel_variant(variant)
    try:
        out = binding.decompress_linear_device(fmt, _dev(torch, data), W, H, status=status)
        torch.cuda.synchronize()
    finally:
        binding.set_kernel_variant(0)
    got = out.cpu().numpy()
    assert np.array_equal(got, want), _first_diff(got, want, 16 * fmt.pixel_bytes)
    assert bool(status.item() == 0) == ok_o


