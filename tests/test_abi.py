"""CPU tests of the C-ABI boundary: the shared library loads without a GPU, exports every symbol
include/gsr.h declares, its host-only entry points work, and argument-contract errors come back as
codes + messages (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(g(?:sr|ab|ls)_[a-z0-9_]+)\s*\(", txt)))


def _header_abi(header):
    return int(re.search(r"#define\s+G[A-Z]+_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", header)).read()).group(1))


def test_gsr_library_exports_every_declared_symbol():
    from gaussianavatars_amd import _lib

    lib = _lib.gsr()
    names = _declared("gsr.h")
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"include/gsr.h declares {n} but libgsr_hip.so does not export it"
        assert n in _lib.GSR_SYMBOLS, f"{n} has no ctypes prototype in _lib.GSR_SYMBOLS"
    assert lib.gsr_abi_version() == _lib.GSR_ABI_VERSION == _header_abi("gsr.h")


def test_layouts_are_disjoint_and_aligned():
    from gaussianavatars_amd import _lib

    lib = _lib.gsr()
    gl, bl, il = _lib.GsrGeomLayout(), _lib.GsrBinningLayout(), _lib.GsrImageLayout()
    assert lib.gsr_geom_layout(100_000, C.byref(gl)) == 0
    assert lib.gsr_binning_layout(3_000_000, 550, 802, 100_000, 0, C.byref(bl)) == 0
    assert lib.gsr_image_layout(550, 802, C.byref(il)) == 0
    P, cap, tiles, HW = 100_000, 3_000_000, 35 * 51, 550 * 802
    segs = [(gl.depths, 4 * P), (gl.grec, 48 * P), (gl.cov3D, 24 * P),
            (gl.rect, 8 * P), (gl.tiles_touched, 4 * P), (gl.clamped, P), (gl.visible, P), (gl.brec, 48 * P), (gl.acc64, 80 * P), (gl.acc, 48 * P)]
    _check(segs, gl.total)
    # parity modes: the rank path (tile lists ordered by global depth rank), reference-format lists materialised
    assert bl.path == 0 and bl.nb == 512 and bl.nbands == 1 and bl.band_rows == 51 and bl.obs == bl.bandcnt
    nblk = (P + 255) // 256
    rank_segs = [(bl.qpos, 16 * cap), (bl.qcount, 16 * tiles), (bl.qstart, 16 * tiles),
                 (bl.ranges, 8 * tiles), (bl.tile_count, 4 * tiles), (bl.tile_start, 4 * tiles), (bl.tile_cursor, 4 * tiles), (bl.tile_order, 4 * tiles),
                 (bl.block_hist, 4 * 256 * tiles), (bl.dkeys, 8 * P), (bl.dtmp, 8 * P), (bl.bcount, 4 * bl.nb), (bl.bstart, 4 * bl.nb),
                 (bl.bcursor, 4 * bl.nb), (bl.bhist, 4 * 256 * bl.nb), (bl.ranks, 8 * cap), (bl.rank, 4 * P), (bl.srect, 8 * P), (bl.sspan, 32 * P), (bl.pstat, 8 * nblk), (bl.tdesc, 16 * tiles)]
    _check(rank_segs + [(bl.keys, 8 * cap), (bl.point_list, 4 * cap), (bl.qlist, 16 * cap)], bl.total)
    # production settings up to 262144 splats: the same path without the reference-format lists
    assert lib.gsr_binning_layout(cap, 550, 802, P, 3, C.byref(bl)) == 0 and bl.path == 0
    assert bl.keys == bl.point_list and bl.qlist == bl.qpos
    _check([(bl.qpos, 16 * cap), (bl.ranks, 8 * cap), (bl.point_list, 4 * cap), (bl.rank, 4 * P), (bl.srect, 8 * P), (bl.pstat, 8 * nblk)], bl.total)
    # round 1's per-tile bitonic sort (tile_culling 5, A/B runs)
    assert lib.gsr_binning_layout(cap, 550, 802, P, 5, C.byref(bl)) == 0 and bl.path == 2
    segs = [(bl.keys, 8 * cap), (bl.point_list, 4 * cap), (bl.qpos, 16 * cap), (bl.qcount, 16 * tiles), (bl.qstart, 16 * tiles),
            (bl.ranges, 8 * tiles), (bl.tile_count, 4 * tiles), (bl.tile_start, 4 * tiles), (bl.tile_cursor, 4 * tiles), (bl.tile_order, 4 * tiles)]
    _check(segs, bl.total)
    # production: depth-ordered scatter into the quadrant streams (capacity counts stream entries)
    capq = 4_000_000
    assert lib.gsr_binning_layout(capq, 550, 802, P, 1, C.byref(bl)) == 0 and bl.path == 0      # up to 262144 splats: the rank path
    assert lib.gsr_binning_layout(capq, 550, 802, 200_000, 1, C.byref(bl)) == 0 and bl.path == 0
    assert lib.gsr_binning_layout(capq, 550, 802, 300_000, 1, C.byref(bl)) == 0 and bl.path == 1
    assert lib.gsr_binning_layout(capq, 550, 802, P, 4, C.byref(bl)) == 0                        # 4: whenever it applies
    assert bl.path == 1 and bl.chunks == 782 and bl.nb == 256 and (P + bl.chunks - 1) // bl.chunks <= 255
    Q = 4 * tiles
    segs = [(bl.qpos, 4 * capq), (bl.qcount, 4 * Q), (bl.qstart, 4 * Q), (bl.tile_order, 4 * tiles), (bl.dkeys, 8 * P), (bl.dtmp, 8 * P),
            (bl.order, 4 * P), (bl.bcount, 4 * bl.nb), (bl.bstart, 4 * bl.nb), (bl.bcursor, 4 * bl.nb), (bl.border, 4 * bl.nb),
            (bl.qhist, bl.chunks * Q), (bl.qprefix, 4 * bl.chunks * Q), (bl.qmask, 16 * P)]
    _check(segs, bl.total)
    # grids beyond 16384 quadrants and splat counts beyond 255 per chunk do not take the scatter: the rank path at any size
    # (one tile bitmap up to 262144 splats; beyond, ranks per band of tile rows); round 1's per-tile sort only on request (tile_culling 5)
    assert lib.gsr_binning_layout(capq, 1600, 1100, 250_000, 1, C.byref(bl)) == 0 and bl.path == 0 and bl.nbands == 1
    assert lib.gsr_binning_layout(capq, 1600, 1100, 2_000_000, 1, C.byref(bl)) == 0 and bl.path == 0 and bl.nb == 4096
    Pb, tb = 2_000_000, 100 * 69
    assert bl.band_rows == 3 and bl.nbands == 23          # 69 tile rows in bands of ceil(69 / 24)
    _check([(bl.qpos, 16 * capq), (bl.ranks, 8 * capq), (bl.point_list, 4 * capq), (bl.rank, 16 * Pb), (bl.rank_over, 4 * 23 * Pb), (bl.obs, 8 * Pb),
            (bl.bandcnt, 4 * 23 * ((Pb + 255) // 256)), (bl.srect, 8 * Pb), (bl.sspan, 32 * Pb), (bl.tdesc, 16 * tb)], bl.total)
    assert lib.gsr_binning_layout(capq, 550, 802, P, 6, C.byref(bl)) == 0 and bl.path == 0     # 6: the bands at any splat count (tests)
    assert bl.band_rows == 3 and bl.nbands == 17 and bl.rank + 16 * P <= bl.rank_over and bl.rank_over + 4 * P * 17 <= bl.srect
    assert lib.gsr_binning_layout(capq, 550, 802, 2_000_000, 1, C.byref(bl)) == 0 and bl.path == 0
    assert lib.gsr_binning_layout(capq, 1600, 1100, 2_000_000, 5, C.byref(bl)) == 0 and bl.path == 2
    _check([(il.final_T, 4 * HW), (il.n_contrib, 4 * HW), (il.n_contrib_q, 4 * HW), (il.c_final, 12 * HW), (il.ck, 144 * HW)], il.total)
    assert lib.gsr_geom_layout(-1, C.byref(gl)) < 0 and b"bad arguments" in lib.gsr_last_error()


def _check(segs, total):
    segs = sorted(segs)
    for (o, n), (o2, _) in zip(segs, segs[1:]):
        assert o % 16 == 0 and o + n <= o2
    assert segs[-1][0] + segs[-1][1] <= total


def test_forward_argument_contract_without_gpu():
    """The two reference error messages (SURVEY.md 8(b) Errors) are produced by the C ABI before any
    device work; nothing here touches a GPU."""
    from gaussianavatars_amd import _lib

    lib = _lib.gsr()
    s = _lib.GsrSettings()
    s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier, s.sh_degree = 32, 32, 0.5, 0.5, 1.0, 0
    n = C.c_int64(0)
    one = C.c_void_p(16)  # never dereferenced: validation fails first
    assert lib.gsr_forward(C.byref(s), 4, 1, one, one, None, one, one, one, None, one, one, one, one, 64, one, C.byref(n), None) < 0
    assert b"device pointers" in lib.gsr_last_error()
    s.bg = s.viewmatrix = s.projmatrix = s.campos = 16
    rc = lib.gsr_forward(C.byref(s), 4, 1, one, one, one, one, one, one, None, one, one, one, one, 64, one, C.byref(n), None)
    assert rc < 0 and b"SHs or precomputed colors" in lib.gsr_last_error()
    rc = lib.gsr_forward(C.byref(s), 4, 1, one, one, None, one, one, None, None, one, one, one, one, 64, one, C.byref(n), None)
    assert rc < 0 and b"scale/rotation pair" in lib.gsr_last_error()
    s.sh_degree = 2
    rc = lib.gsr_forward(C.byref(s), 4, 4, one, one, None, one, one, one, None, one, one, one, one, 64, one, C.byref(n), None)
    assert rc < 0 and b"coefficients" in lib.gsr_last_error()
    s.sh_degree = 0   # the splat-count ceiling of the 32-bit record offsets
    rc = lib.gsr_forward(C.byref(s), 89_478_486, 1, one, one, None, one, one, one, None, one, one, one, one, 64, one, C.byref(n), None)
    assert rc < 0 and b"89478485" in lib.gsr_last_error()


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: no file of the product package may reference it."""
    pkg = os.path.join(ROOT, "gaussianavatars_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|#include\s+\".*oracle", txt, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_rasterizer_refuses_cpu_tensors():
    import torch

    from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    z = torch.zeros
    rs = GaussianRasterizationSettings(16, 16, 0.5, 0.5, z(3), 1.0, torch.eye(4), torch.eye(4), 0, z(3), False, False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        GaussianRasterizer(rs)(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 1, 3), scales=z(2, 3), rotations=z(2, 4))


def test_deferred_count_slots_are_a_finite_pool():
    """rasterizer.deferred_count hands every forward inside it one of the GSR_COUNT_SLOTS persistent slots of include/gsr.h
    (GsrSettings.deferred_count); release() returns them; running out is an error, not a silent reuse."""
    import pytest

    from gaussianavatars_amd import _lib
    from gaussianavatars_amd import rasterizer as R

    assert _lib.GSR_COUNT_SLOTS == 128 and "deferred_count" in [f[0] for f in _lib.GsrSettings._fields_]
    free0 = len(R._free_slots)
    with R.deferred_count(1000) as d:
        assert R._deferred is d and d.capacity == 1 << 16        # capacities come in 64 K quanta
        taken = [d.take() for _ in range(3)]
    assert R._deferred is None and len(set(taken)) == 3 and len(R._free_slots) == free0 - 3
    many = R._Deferred(1)
    for _ in range(free0 - 3):
        many.take()
    with pytest.raises(RuntimeError, match="slots are in use"):
        many.take()
    many.release()
    d.release()
    assert sorted(R._free_slots) == list(range(128)) and len(R._free_slots) == free0


def _header_struct_fields(header, name):
    txt = open(os.path.join(ROOT, "include", header)).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            m = re.match(r"(.*?)(\w+)$", decl, flags=re.S)
            out.append((m.group(2), " ".join(m.group(1).split())))
    return out


def test_integration_stub_matches_the_header():
    """INTEGRATION.md section 2 prints the ctypes stub a maintainer would copy.  A stale stub hands the library a short GsrSettings (garbage in
    the trailing switches): the field list of include/gsr.h, of the shipped ctypes structure and of the printed stub must be the same,
    in order and in type, and the calls the stub makes must pass as many arguments as the prototypes take."""
    from gaussianavatars_amd import _lib

    ctype_of = {"int32_t": C.c_int32, "float": C.c_float, "const float*": C.c_void_p}
    hdr = _header_struct_fields("gsr.h", "GsrSettings")
    want = [(n, ctype_of[t]) for n, t in hdr]
    shipped = [(n, t) for n, t in _lib.GsrSettings._fields_]
    assert [n for n, _ in shipped] == [n for n, _ in want]
    for (n, t), (_, w) in zip(shipped, want):
        assert C.sizeof(t) == C.sizeof(w), n
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes as C, torch\n.*?)```", doc, flags=re.S).group(1)
    cls = re.search(r"(class GsrSettings\(C\.Structure\):.*?\n)\ndef ", block, flags=re.S).group(1)
    ns = {"C": C}
    exec(cls, ns)
    stub = ns["GsrSettings"]._fields_
    assert [n for n, _ in stub] == [n for n, _ in want], "INTEGRATION.md prints a GsrSettings that is not include/gsr.h's"
    assert C.sizeof(ns["GsrSettings"]) == C.sizeof(_lib.GsrSettings)
    for (n, t), (_, w) in zip(stub, want):
        assert C.sizeof(t) == C.sizeof(w), n
    assert "gsr_abi_version() == %d" % _header_abi("gsr.h") in block
    # the positional GsrSettings(...) construction fills every field
    ctor = re.search(r"s = GsrSettings\((.*?)\)\n    gl, bl", block, flags=re.S).group(1)
    depth, nargs = 0, 1
    for ch in ctor:
        depth += ch in "([" 
        depth -= ch in ")]"
        nargs += ch == "," and depth == 0
    assert nargs == len(want)
    # argument counts of the calls against the prototypes
    for fn in ("gsr_geom_layout", "gsr_image_layout", "gsr_binning_layout", "gsr_forward"):
        call = re.search(r"lib\.%s\((.*?)\)\n" % fn, block, flags=re.S)
        if fn == "gsr_forward":
            call = re.search(r"lib\.gsr_forward\((.*?)\)\n        if rc == 1", block, flags=re.S)
        if fn == "gsr_geom_layout":
            call = re.search(r"lib\.gsr_geom_layout\((.*?)\);", block, flags=re.S)
        args, depth, n = call.group(1), 0, 1
        for ch in args:
            depth += ch in "(["
            depth -= ch in ")]"
            n += ch == "," and depth == 0
        assert n == len(_lib.GSR_SYMBOLS[fn][1]), f"{fn}: the stub passes {n} arguments, the prototype takes {len(_lib.GSR_SYMBOLS[fn][1])}"


def test_compiled_host_loads_without_a_gpu_and_refuses_host_tensors():
    """gaussianavatars_amd/gaa_host.so (csrc/gaa_host.cpp, built by `make host` / __graft_entry__.build()): the compiled host side of the
    autograd nodes maps on a box without a GPU, resolves every C-ABI entry it launches through from the three libraries (init() raises on a
    missing symbol or an ABI version other than the headers'), and -- like the Python twins -- has no CPU path: host tensors are an error."""
    import pytest
    import torch

    from gaussianavatars_amd import _host, _lib

    H = _host.load()
    assert (H.GSR_ABI, H.GAB_ABI, H.GLS_ABI) == (_lib.GSR_ABI_VERSION, _lib.GAB_ABI_VERSION, _lib.GLS_ABI_VERSION)
    assert (H.GSR_ABI, H.GAB_ABI, H.GLS_ABI) == (_header_abi("gsr.h"), _header_abi("gab.h"), _header_abi("gls.h"))
    for name in ("init", "make_mesh_plan", "mesh_frames", "rasterize_bound", "l1_loss", "l1_ssim", "set_unit_seed", "l1_emit_state"):
        assert callable(getattr(H, name)), name
    a, b = torch.rand(3, 4, 5), torch.rand(3, 4, 5)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        H.l1_loss(a, b, -1, False)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        H.l1_ssim(a, b)
    # the switch between the two host sides
    prev = _host.set_enabled(False)
    try:
        assert _host.get() is None
    finally:
        _host.set_enabled(prev)
    assert (_host.get() is H) == prev
