"""Parity of the rejected A/B kernels (profiles/AB_RECORD.md) -- the north_star's own kernel shape (a wave owns a tile of 4x4 blocks, lane =
texel row, LDS staging: variant 1) among them.  They are NOT part of the product library: `make lib-ab` (run by
__graft_entry__.build()) builds them into tests/ab_build/libdetexhip_ab.so, and this module loads THAT build with its own ctypes.CDLL,
beside the product library the rest of the suite uses -- nothing in the environment is needed.  (DETEXHIP_LIB=<another A/B build> is
honoured when it names one.)"""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as ol
from detex_amd import formats as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ENV = os.environ.get("DETEXHIP_LIB", "")
AB_LIB = _ENV if "_ab" in os.path.basename(_ENV) else os.path.join(ROOT, "tests", "ab_build", "libdetexhip_ab.so")

pytestmark = [pytest.mark.gpu]


class _AbBinding:
    """the three entry points these tests need, over the A/B build (same C ABI as the product: include/detexhip.h)"""

    def __init__(self, path):
        import torch  # noqa: F401  (torch's bundled HIP runtime first: detex_amd/binding.py has the reason)
        assert os.path.exists(path), "%s is missing: `make lib-ab` (part of __graft_entry__.build())" % path
        self.lib = lib = ctypes.CDLL(path)
        lib.detexGetErrorMessage.restype = ctypes.c_char_p
        lib.detexhipSetKernelVariant.argtypes = [ctypes.c_int]
        lib.detexhipGetKernelVariant.restype = ctypes.c_int
        vp = ctypes.c_void_p
        lib.detexhipDecompressTextureLinearDevice.argtypes = [ctypes.c_uint32, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t,
                                                              ctypes.c_uint32, vp, vp]
        lib.detexhipDecompressTextureTiledDevice.argtypes = [ctypes.c_uint32, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_uint32, vp, vp]

    def set_kernel_variant(self, v):
        self.lib.detexhipSetKernelVariant(int(v))
        assert self.lib.detexhipGetKernelVariant() == int(v), "this build does not have variant %d: not an A/B build?" % v

    def _stream(self):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def decompress_linear_device(self, fmt, blocks, width, height, status=None):
        import torch
        px = fmt.pixel_bytes
        out = torch.empty(width * height * px, dtype=torch.uint8, device=blocks.device)
        rc = self.lib.detexhipDecompressTextureLinearDevice(fmt.texture_format, blocks.data_ptr(), width, height, (width + 3) // 4, (height + 3) // 4, out.data_ptr(),
                                                            width * px, F.native_pixel_format(fmt), self._stream(), None if status is None else status.data_ptr())
        assert rc == 0, self.lib.detexGetErrorMessage()
        return out

    def decompress_tiled_device(self, fmt, blocks, wb, hb, status=None):
        import torch
        out = torch.empty(wb * hb * 16 * fmt.pixel_bytes, dtype=torch.uint8, device=blocks.device)
        rc = self.lib.detexhipDecompressTextureTiledDevice(fmt.texture_format, blocks.data_ptr(), wb, hb, out.data_ptr(), F.native_pixel_format(fmt), self._stream(),
                                                           None if status is None else status.data_ptr())
        assert rc == 0, self.lib.detexGetErrorMessage()
        return out


@pytest.fixture(scope="module")
def binding(torch_cuda):
    return _AbBinding(AB_LIB)


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


def test_bc1_tile4x4_variant_matches(binding, torch_cuda, oracle):
    """the north_star tile-shape variant (A/B only) decodes identically"""
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


@pytest.mark.parametrize("name,variant", [("BPTC", 3), ("BPTC", 4), ("BPTC", 5), ("BPTC_FLOAT", 3), ("BPTC_SIGNED_FLOAT", 3), ("BC1", 2), ("BPTC_FLOAT", 2),
                                          ("BC1", 8), ("BC3", 8), ("ETC2", 8), ("BPTC", 8), ("EAC_RG11", 8), ("BC1", 9), ("BC1A", 9), ("BC1", 10), ("BC3", 10),
                                          ("BPTC_FLOAT", 11), ("BPTC_SIGNED_FLOAT", 11)])
def test_alternative_decoder_variants_match(name, variant, binding, torch_cuda, oracle, forced_vectors):
    """the A/B decoder implementations and kernel shapes (profiles/AB_RECORD.md; 8-11: the store-shape kernels of round 6, tools/ab/kernels_store_shape.h)
    decode identically, forced classes included"""
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


