"""A whole frame step as ONE hipGraph launch.

The frame loop of the hot path (select_mesh_by_timestep -> render -> loss -> backward) is ~30 kernel launches driven by ~0.3 ms
of Python per step.  On a GPU that needs 0.4 ms for the kernels that is fine while the process runs next to its GPU and alone;
it stops being fine on a slower, busier or unpinned host.  `GraphedStep` records the step once -- stream capture sees every
launch the three native libraries make, they are plain hipLaunchKernel calls on torch's current stream -- and replays it with
one hipGraphLaunch: the host cost of a step becomes a few microseconds.

What makes the step capturable:
  * the rasterizer forward does not wait for the frame's instance count (`rasterizer.deferred_count`, include/gsr.h
    GsrSettings.deferred_count): the binning buffer is over-allocated (`headroom` x the count of the warm-up frames; 288 GB of
    HBM make that free), the count goes to a persistent slot, and `check()` reads it after the fact.  A frame whose count
    exceeds the capacity skips its binning and blend kernels (nothing is read or written out of bounds): `check()` raises
    CapacityOverflow for it and `recapture()` doubles the headroom;
  * every input that changes from frame to frame lives in a STATIC device tensor the caller refills before `replay()`:
    camera matrices (the rasterizer reads them through pointers), the FLAME parameters of the frame (`FlameRowFeeder`:
    one-row tables fed from the packed sequence by ONE device-to-device copy), the target image;
  * scalars passed by value (image size, tan(fov), SH degree, splat count) are part of the recording: a graph is valid for one
    model size and one camera geometry.  Densification changes the model: capture again afterwards.

Gradients: parameters' .grad must be None when the step is captured; the recorded backward then allocates them inside the
graph's memory pool and every replay rewrites the same tensors in place (torch's whole-network capture rules).

Autograd state: a recording must not meet pieces of an OLDER autograd graph.  torch ties a leaf's AccumulateGrad node to the
stream it was created on and keeps the node alive as long as any graph refers to it; a mesh-bound model holds such a graph
between frames (`model.verts`, `model.face_center`, ... are outputs of the mesh node of the previous frame).  If that frame ran
on another stream, the recorded backward would hop to that stream in the middle of the capture (observed: a segmentation fault
inside hipStreamEndCapture).  So the warm-up frames and the recording share ONE stream, and `before_capture` -- called before
every warm-up frame and before the recording -- should drop what the model keeps from the previous frame
(`release_mesh(model)`) along with the gradients.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import rasterizer as _R

__all__ = ["GraphedStep", "CapacityOverflow", "FlameRowFeeder", "release_mesh", "shared_lane_model", "accumulate_lane_grads", "LEAF_NAMES"]


def release_mesh(model) -> None:
    """Drops the per-frame mesh tensors select_mesh_by_timestep left on the model (scene/flame_gaussian_model.py:137-154) and
    with them the autograd graph of the previous frame."""
    for k in ("verts", "verts_cano", "face_center", "face_orien_mat", "face_scaling", "face_orien_quat"):
        if hasattr(model, k):
            setattr(model, k, None)


LEAF_NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def shared_lane_model(model):
    """A second handle on `model` for another frame lane (a recorded step on its own stream): the six leaf parameters are new leaf tensors
    over the SAME storage (`detach()` shares memory: the lanes read one set of splats, 236 B per splat once per GPU, not once per lane), so
    each lane's backward writes its OWN `.grad` tensors; the FLAME tables are the lane's own clones (a recorded lane feeds its frame into
    static one-row tables, FlameRowFeeder) and the per-frame mesh tensors start empty.  Everything else (rig, binding, CSR, counters) is
    shared by reference.  `accumulate_lane_grads` adds the lanes' gradients up: S lanes x K frames = gradient accumulation over S K frames,
    after which ONE optimiser step on `model`'s leaves moves every lane (same storage)."""
    import copy

    lane = copy.copy(model)                      # shallow: attributes by reference
    for name in LEAF_NAMES:
        p = getattr(model, name)
        setattr(lane, name, p.detach().requires_grad_(p.requires_grad))
    fp = getattr(model, "flame_param", None)
    if fp is not None:
        lane.flame_param = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in fp.items()}
    release_mesh(lane)
    return lane


def accumulate_lane_grads(model, lanes, flame: bool = True) -> None:
    """model.<leaf>.grad (+)= sum over `lanes` of their <leaf>.grad, on the device, in lane order (torch._foreach_add_: one fused launch per
    lane); with `flame`, the same for the FLAME tables that require gradients.  Call it on a stream that has waited for the lanes' streams.
    The sum order is fixed, so with the rasterizer's deterministic backward the result is the bits of the sequential accumulation
    loss_0.backward(); loss_1.backward(); ... on one model."""
    names = list(LEAF_NAMES)
    targets = [getattr(model, n) for n in names]
    if flame and getattr(model, "flame_param", None) is not None:
        keys = [k for k, v in model.flame_param.items() if v.requires_grad]
        targets += [model.flame_param[k] for k in keys]
    else:
        keys = []
    with torch.no_grad():
        for lane in lanes:
            srcs = [getattr(lane, n).grad for n in names] + [lane.flame_param[k].grad for k in keys]
            dst, add = [], []
            for t, g in zip(targets, srcs):
                if g is None:
                    continue
                if t.grad is None:
                    t.grad = g.detach().clone()
                else:
                    dst.append(t.grad)
                    add.append(g)
            if dst:
                torch._foreach_add_(dst, add)


class CapacityOverflow(RuntimeError):
    """A replayed frame produced more tile instances than the recorded binning capacity holds; that frame's image and
    gradients are not valid."""


class GraphedStep:
    """step = GraphedStep(fn); step.replay() -> fn's recorded return value (static tensors), step.check().

    `fn()` runs one frame step on static tensors (forward, loss, backward) and returns a tensor or a tuple of tensors.
    It is run `warmup` times eagerly on a side stream first -- every lazy initialisation (library loads, rig preparation,
    binding CSR, LDS attributes) happens there, and the rasterizer learns the frame's instance count -- then once more
    under stream capture.  `before_capture()` (optional) runs right before the recording, e.g. to reset .grad to None."""

    def __init__(self, fn: Callable[[], object], warmup: int = 3, headroom: float = 4.0, before_capture: Optional[Callable[[], None]] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs the GPU (hipGraph capture); there is no CPU path")
        self.fn, self.before_capture = fn, before_capture
        self.headroom = float(headroom)
        self.warmup = max(1, int(warmup))
        self.replays = 0
        self._state = None
        self._stream = torch.cuda.Stream()   # the warm-up frames and the recording run here (see the module docstring)
        self._capture()

    def _capture(self):
        side = self._stream
        side.wait_stream(torch.cuda.current_stream())
        need = 0
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                if self.before_capture is not None:
                    self.before_capture()
                _R._forward_peak[0] = 0
                self.fn()
                need = max(need, int(_R._forward_peak[0]))   # the largest count of ANY rasterizer forward the step issued
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self._state is not None:
            self._state.release()
        self._state = _R._Deferred(int(self.headroom * max(need, 1)) + 1)
        self.warm_instances = need
        if self.before_capture is not None:
            self.before_capture()
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph, stream=side):
                with _R.deferred_count(self._state):
                    self.out = self.fn()
            if not self._state.slots:
                raise RuntimeError("GraphedStep: fn() issued no rasterizer forward")
        except BaseException:
            self._state.release()   # the count slots go back to the pool whatever went wrong in the recording
            self._state = None
            raise

    @property
    def capacity(self) -> int:
        return self._state.capacity

    def replay(self):
        """One hipGraphLaunch on the current stream; returns the recorded outputs (valid once the stream gets there)."""
        self.graph.replay()
        self.replays += 1
        return self.out

    def instances(self) -> list:
        """Instance count of the newest frame each recorded rasterizer forward has finished (-1: none yet).  No synchronisation:
        call after a sync (or an event) when the count of a particular replay is wanted."""
        return self._state.counts()

    def check(self) -> None:
        """Raises CapacityOverflow when ANY replay since the recording (or since the previous check that raised) did not fit the recorded
        capacity: the device leaves a sticky mark per count slot (include/gsr.h: gsr_count_slot_overflow), so one call after a run of
        replays -- once their work has finished -- covers all of them.  The mark is cleared when it is reported."""
        worst = max(max(self.instances()), self._state.overflow(reset=True))
        if worst > self.capacity:
            raise CapacityOverflow(f"{worst} tile instances exceed the recorded binning capacity {self.capacity} "
                                   f"(captured at {self.warm_instances} with headroom {self.headroom}): that frame is not valid; recapture()")

    def recapture(self, headroom: Optional[float] = None) -> None:
        """Record again (after densification, a new camera geometry, or an overflow: the headroom doubles unless given)."""
        self.headroom = float(headroom) if headroom is not None else 2.0 * self.headroom
        torch.cuda.synchronize()
        self._capture()

    def close(self) -> None:
        if self._state is not None:
            torch.cuda.synchronize()   # a replay still in flight would post into a slot its next owner already holds
            self._state.release()
            self._state = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FlameRowFeeder:
    """Static one-row FLAME parameter tables for a recorded step.

    select_mesh_by_timestep(t) addresses row t of the (T, k) tables of flame_param (scene/flame_gaussian_model.py:117-135);
    a recording would freeze t.  The feeder packs the per-timestep tables into one (T, sum k) tensor, hands the model
    one-row VIEWS of a single static row -- `model.flame_param = feeder.static_param`, `select_mesh_by_timestep(0)` inside the
    recorded step -- and `feed(t)` refreshes that row with one device-to-device copy before a replay."""

    ROWS = ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")

    def __init__(self, flame_param: Dict[str, torch.Tensor], requires_grad: bool = False):
        tabs = [flame_param[k].detach() for k in self.ROWS]
        self.T = int(tabs[0].shape[0])
        self.packed = torch.cat([t.reshape(self.T, -1).float() for t in tabs], dim=1).contiguous()
        self.row = self.packed[:1].clone()
        self.schedule = self.cursor = None   # set_schedule(): the device-side order of the frames for feed_next()
        # (shape, static_offset and dynamic_offset pass through: FlameHead.forward accepts dynamic_offset and never reads it,
        #  flame_model/flame.py:498)
        self.static_param = {k: v for k, v in flame_param.items() if k not in self.ROWS}
        off = 0
        for k, t in zip(self.ROWS, tabs):
            w = int(t.reshape(self.T, -1).shape[1])
            view = self.row[:, off:off + w]
            self.static_param[k] = view.requires_grad_(True) if requires_grad else view
            off += w

    def feed(self, t: int) -> None:
        t = int(t) % self.T
        with torch.no_grad():
            self.row.copy_(self.packed[t:t + 1])

    # ---- the feed as a kernel INSIDE a recording: K frames per graph, each preceded by feed_next() -------------------------
    def set_schedule(self, frames, start: int = 0) -> None:
        """The timesteps successive feed_next() calls deliver, cyclically (a DEVICE table: replays of a recording that contains
        feed_next() walk it on their own), and the position the next one starts from."""
        dev = self.packed.device
        self.schedule = torch.as_tensor([int(f) % self.T for f in frames], dtype=torch.int32, device=dev)
        self.cursor = torch.full((1,), int(start), dtype=torch.int32, device=dev)

    def seek(self, position: int) -> None:
        """Next feed_next() delivers schedule[position % len] (host call between replays; one fill kernel)."""
        if getattr(self, "cursor", None) is None:
            raise RuntimeError("FlameRowFeeder.seek(): call set_schedule(frames) first")
        self.cursor.fill_(int(position))

    def feed_next(self) -> None:
        """row <- packed[schedule[cursor]]; cursor = (cursor + 1) mod len(schedule) -- one kernel on the current stream (include/gab.h: gab_feed_row), capturable."""
        import ctypes as C

        from . import _lib

        if getattr(self, "schedule", None) is None:
            raise RuntimeError("FlameRowFeeder.feed_next(): call set_schedule(frames) first (the order of the frames lives on the device)")
        dev = self.packed.device
        if dev.type != "cuda":   # host tensors (the gloo dry run of the frame-parallel plumbing): the same walk, in Python
            n = int(self.schedule.numel())
            c = int(self.cursor.item()) % n
            self.feed(int(self.schedule[c].item()))
            self.cursor.fill_((c + 1) % n)
            return
        with _lib.on_device(dev):
            rc = _lib.gab().gab_feed_row(C.c_void_p(self.packed.data_ptr()), self.T, int(self.packed.shape[1]), C.c_void_p(self.schedule.data_ptr()),
                                         int(self.schedule.numel()), C.c_void_p(self.cursor.data_ptr()), C.c_void_p(self.row.data_ptr()),
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gab_feed_row: {_lib.gab().gab_last_error().decode()}")
