"""The drop-in boundary exercised by a COMPILED C PROGRAM, not Python: tests/c_client/detex_client.c is built with gcc against a
detex.h and linked with -ldetexhip in place of -ldetex (`make c-client`, part of __graft_entry__.build()); it makes the
reference's own call sequence (validate.c:135,199-209: detexLoadKTXFile -> detexDecompressTextureLinear) in a fresh process that
contains neither Python nor torch, and prints the sha256 of what it decoded.  Two binaries from the one source: `detex_client`
(this repository's include/detex.h) and, where the build container had the reference's sources, `detex_client_refhdr`, compiled
against the REFERENCE's own detex.h -- the program a libdetex user already has, re-linked."""
import hashlib
import os
import subprocess

import pytest

import oracle_lib as ol
from detex_amd import formats as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENTS = [os.path.join(ROOT, "tests", "c_client", n) for n in ("detex_client", "detex_client_refhdr")]
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _clean_env():
    """the environment of a C user: no Python paths, nothing preloaded"""
    return {k: v for k, v in os.environ.items() if not k.startswith(("PYTHON", "LD_PRELOAD"))}


def _clients():
    return [c for c in CLIENTS if os.path.exists(c)]


@pytest.fixture(scope="module")
def built():
    # (the GPU box gets the binaries __graft_entry__.build() made, not build/ with the library's object files: `make` there would recompile the
    # product library under the running test -- only where the objects are, or where a binary is missing)
    if os.path.isdir(os.path.join(ROOT, "build", "obj")) or not os.path.exists(CLIENTS[0]):
        subprocess.check_call(["make", "-s", "-C", ROOT, "c-client"])
    assert os.path.exists(CLIENTS[0])
    if os.path.exists("/root/reference/detex.h"):
        assert os.path.exists(CLIENTS[1]), "the build container has the reference's header: the second client must exist"
    return _clients()


def test_c_client_builds_links_and_hashes(built):
    """no GPU needed: the client links against the library alone (no torch in its dependency closure, libamdhip64 from the ROCm
    installation), and its sha256 is FIPS 180-4's"""
    for exe in built:
        ldd = subprocess.run(["ldd", exe], capture_output=True, text=True, env=_clean_env()).stdout
        assert "libdetexhip.so" in ldd and "not found" not in ldd, ldd
        hip = [l for l in ldd.splitlines() if "libamdhip64" in l]
        assert hip and "torch" not in hip[0], ldd
        assert "torch" not in ldd and "python" not in ldd.lower()
        out = subprocess.run([exe, "--sha256-selftest"], capture_output=True, text=True, env=_clean_env(), timeout=60).stdout.split()
        assert out[1] == hashlib.sha256(b"abc").hexdigest() and out[3] == hashlib.sha256(b"a" * 1000000).hexdigest()


def _parse(line):
    fields = dict(f.split("=", 1) for f in line.split()[1:] if "=" in f)
    return line.split()[0], fields


@pytest.mark.gpu
def test_c_client_decodes_the_fixtures_and_a_full_size_stream(built, golden_json):
    """on the GPU box: the 17 bundled fixtures file -> pixels, and the 8192^2 BC1 stream U, through the compiled client; every digest
    equals the compiled reference's (tests/golden/fixtures.json, digests_8192.json)"""
    fx = golden_json("fixtures.json")
    dg = golden_json("digests_8192.json")["streams"]
    files = [os.path.join(GOLDEN, f.fixture) for f in F.FORMATS if f.fixture]
    for exe in built:
        ldd = subprocess.run(["ldd", exe], capture_output=True, text=True, env=_clean_env()).stdout
        print(os.path.basename(exe), "resolves:", *[l.strip() for l in ldd.splitlines() if "amdhip" in l or "detexhip" in l], sep="\n  ")
        r = subprocess.run([exe] + files, capture_output=True, text=True, env=_clean_env(), timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
        lines = r.stdout.strip().splitlines()
        assert len(lines) == len(files) == 17
        for line in lines:
            path, got = _parse(line)
            fmt = next(f for f in F.FORMATS if f.fixture == os.path.basename(path))
            want = fx[fmt.name]["0x%04X" % F.native_pixel_format(fmt)]
            assert int(got["format"], 16) == fmt.texture_format
            assert got["sha256"] == want["sha256"] and (got["ok"] == "1") == want["ok"], (os.path.basename(exe), fmt.name)
        # a full-size texture through the host tier: 8192^2, stream U (one with invalid blocks) -- with the upload beside the download (32+ MiB of
        # blocks: the library's helper thread) and, DETEXHIP_HOST_DUPLEX=0, without
        for name, duplex in (("BC1", "1"), ("BPTC", "1"), ("BPTC", "0")):
            fmt = F.BY_NAME[name]
            seed = ol.STREAM_SEED_BASE + ol.STREAM_SEED_K[name]
            r = subprocess.run([exe, "--stream", "0x%08X" % fmt.texture_format, str(fmt.block_bytes), "0x%X" % seed, "8192", "8192"],
                               capture_output=True, text=True, env=dict(_clean_env(), DETEXHIP_HOST_DUPLEX=duplex), timeout=600)
            assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
            _, got = _parse(r.stdout.strip().splitlines()[-1])
            want = dg["%s/U" % name]
            assert got["sha256"] == want["sha256"] and (got["ok"] == "1") == want["ok"], (os.path.basename(exe), name, duplex)


@pytest.mark.gpu
@pytest.mark.parametrize("idle_us", ["100", "0", "3", "1"])
def test_c_client_back_to_back_small_calls(built, idle_us):
    """2200 one-block calls and 2200 small textures per size in a tight loop from compiled C, another block every call: with the
    resident kernel answering (default idle time), with a launch per call (0), and with an idle time so short that the kernel keeps
    leaving between requests (3 us: every way a request can meet a leaving kernel) -- no wrong or stale answer in any of them"""
    env = dict(_clean_env(), DETEXHIP_RESIDENT_US=idle_us)
    r = subprocess.run([built[0], "--latency"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = {}
    for l in r.stdout.splitlines():
        if l.startswith("latency "):
            rows.update(_parse(l)[1])
    assert rows.get("wrong_results") == "0", r.stdout
    assert all(k in rows for k in ("one_block_us", "64x64_us", "128x128_us", "256x256_us")), r.stdout


@pytest.mark.gpu
def test_c_client_batched_blocks_against_the_leaf_loop(built):
    """the migration path of a per-block client from compiled C (both headers): 1, 1024 and 1048576 independent BC1 / BC7 blocks through ONE
    detexhipDecompressBlocks call == the loop over the leaf function block by block (pixels and ok bytes; n <= 1024), and the call's bool
    == all(ok)"""
    for exe in built:
        r = subprocess.run([exe, "--blocks"], capture_output=True, text=True, env=_clean_env(), timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        rows = [l for l in r.stdout.splitlines() if l.startswith("blocks format=")]
        assert len(rows) == 6 and "blocks wrong_results=0" in r.stdout, r.stdout
        for l in rows:
            assert float(_parse(l)[1]["batched_us"]) > 0, l
        print(os.path.basename(exe)); print(r.stdout)
