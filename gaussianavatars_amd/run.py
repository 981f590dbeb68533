"""Launcher: run an UNCHANGED entry script of the reference (train.py, render.py, fps_benchmark_demo.py,
fps_benchmark_dataset.py, ...) on MI355X.

    cd <reference checkout>
    python -m gaussianavatars_amd.run fps_benchmark_demo.py --point_path media/306/point_cloud.ply
    python -m gaussianavatars_amd.run train.py -s <data> -m <out> --bind_to_mesh

Equivalent to `import gaussianavatars_amd.patch as P; P.patch_reference()` at the top of the script: import shims,
`diff_gaussian_rasterization` -> the HIP rasterizer, the per-frame model methods -> the fused binding kernels, and
`gaussian_renderer.render` -> the mirror with the same signature.  GSR_FAST_RENDER=0 keeps the reference's own render().
One process per GPU: select the device with HIP_VISIBLE_DEVICES (utils/general_utils.py:133 pins cuda:0).
The process is moved onto eight cores next to its GPU first (frame_parallel.pin_host_process: what bench.py measures under; the
un-pinned frame loop is up to a third slower on a two-socket host); GAA_PIN=0 leaves the CPU mask alone.
"""
from __future__ import annotations

import os
import runpy
import sys


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    root = os.path.dirname(script)
    if root not in sys.path:
        sys.path.insert(0, root)
    from . import patch

    info = patch.patch_reference(reference_root=root, fast_render=os.environ.get("GSR_FAST_RENDER", "1") != "0")
    print(f"[gaussianavatars_amd] shims: {', '.join(info['shims']) or 'none'}; fused model methods on "
          f"{', '.join(c.__name__ for c in info['classes'])}; render fast path: {info['render']}; "
          f"host CPUs: {info['pinned_cpus'] if info['pinned_cpus'] else 'unchanged'}; fused loss / statistics: {', '.join(info['loss']) or 'none'}", file=sys.stderr)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
