"""True streaming mode: frames are pushed one pair at a time, the 7-frame SmoothNet window
slides over a ring buffer, TemporalNet features of the previous frame are cached, and every push renders at most one new
frame onto a FIXED canvas.

The reference is only "online" up to the smoothing window: its renderer waits for the whole clip because the canvas is
the bounding box over all frames (test_online_tra.py:106-120).  Here the canvas is fixed when the first window is
complete (bbox of its 7 frames grown by `margin`), or given by the caller; with the offline bbox passed in, the stream
reproduces the offline frames (tests/test_gpu_parity.py::test_online_matches_offline).  Per pushed pair the arithmetic is
exactly the reference's per-frame arithmetic (test_online_tra.py:284-392 with k = t).

Canvas overflow (round 5): the reference would have sized the canvas from ALL frames; here every push also runs a one-wave watcher
(`ss_canvas_watch`) that records, on the device, whether the frame's mesh left the fixed canvas (`clipped_frames`,
`overflow_report()`: read lazily, no sync on the push path).  `grow='never'` (default) keeps the canvas and counts;
`grow='recapture'` re-fixes the canvas (union of the old one and everything seen, plus the margin) and re-captures the graph as soon
as a frame comes within half the margin of an edge -- checked through an asynchronous copy of the watcher state one push later, so
a gradual drift grows the canvas BEFORE anything is cropped (an abrupt jump still crops the frames in between; they are counted).

Once the window is full every push runs the same ~150 small kernels on buffers of fixed size, so the steady state is
captured ONCE into a HIP graph (state lives in static tensors: the rings are shifted, not rotated) and each push is
two input copies + one graph launch: the Python / ctypes launch overhead (~1 ms per pair, more than the kernels'
own time at batch 1) disappears.  `use_graph=False` runs the same code eagerly.
"""
import os
import time

import torch

from . import layers as L
from . import ops, pipeline
from .spatial_network import build_SpatialNet, get_rigid_mesh, get_norm_mesh

WINDOW = pipeline.WINDOW
_warmup = {}


def _new_graph():
    """A CUDAGraph that keeps its hipGraph_t (torch >= 2.5: keep_graph=True) so that `_graph_nodes` can count its nodes."""
    try:
        return torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:
        return torch.cuda.CUDAGraph()


def _graph_nodes(g):
    """Nodes of a captured graph (hipGraphGetNodes on the raw handle), None where the runtime / torch does not expose it.  The
    streaming push is bound by its node count (a node costs >= 4.5 us whatever it does): bench.py prints it."""
    try:
        import ctypes
        raw = g.raw_cuda_graph()
        n = ctypes.c_size_t(0)
        hip = ctypes.CDLL('libamdhip64.so')
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
            return None
        return int(n.value)
    except Exception:
        return None


def _warmup_stream(dev):
    """One capture warm-up stream per device, shared by every stitcher."""
    s = _warmup.get(dev)
    if s is None:
        s = torch.cuda.Stream(dev)
        _warmup[dev] = s
    return s


def _spatial_temporal_heads(spatial, temporal, f64, prev_feat, feat, b, tm_out, chain=False):
    """SpatialNet behind its stage-1 trunk (f64 [2b,45,60,128], view 1 first) and TemporalNet's regressor on the cached /
    current features (prev_feat, feat [2b,45,60,128], view-major) -> (offset_1, offset_2_ref, offset_2_tgt); the temporal
    motions of view 1 / view 2 are written to tm_out = (t1 [b,126], t2 [b,126]).  With layers.QUAD the four regressor heads
    share their launches (layers.run_regressor_quad), else they run as SpatialNet's pair and TemporalNet's own.
    chain: f64 holds b + 1 images of a CHAIN of views (v1 .. v_{b+1}); pair i = (v_i, v_{i+1}) -- the first views are images
    0 .. b-1, the second views images 1 .. b, overlapping slices of one trunk pass (three views: the middle one passes the
    trunks once, threeview:154-343 runs it through both pair passes)."""
    o2 = 1 if chain else b
    if not L.QUAD:
        assert not chain
        off = spatial.forward_features(f64, b, pipeline.LR_H, pipeline.LR_W)
        temporal.motions_from_features(prev_feat, feat, out_slices=[(0, b, tm_out[0]), (b, 2 * b, tm_out[1])])
        return off
    off1 = _heads_a(spatial, f64, b, chain)
    return (off1,) + _heads_b(spatial, temporal, f64, off1, prev_feat, feat, b, tm_out, chain)


def _heads_a(spatial, f64, b, chain=False):
    """First half of the heads (up to the global homography offsets): SpatialNet's stage-2 trunk + contextual correlation +
    regressNet1 -> offset_1 [b,8].  (PipelinedOnlineStitcher cuts the push between the halves.)"""
    o2 = 1 if chain else b
    f32 = L.run_stage2(f64, spatial._prepared()['s2'])
    return spatial.offset1_from_features(f32[:b], f32[o2:o2 + b])


def _heads_b(spatial, temporal, f64, off1, prev_feat, feat, b, tm_out, chain=False):
    """Second half: decomposition, warps, both nets' cost volumes, the four regressor heads in shared launches
    -> (offset_2_ref, offset_2_tgt); temporal motions into tm_out."""
    o2 = 1 if chain else b
    cv_s = spatial.cv_from_offset1(f64[:b], f64[o2:o2 + b], off1, pipeline.LR_H, pipeline.LR_W)
    cv_t = ops.cost_volume(prev_feat, feat, 3, chain=b if chain else 0)      # [2b] volumes, view-major (chain: from b + 1 views stored once)
    off_ref = torch.empty((b, 126), device=f64.device, dtype=torch.float32)
    off_tgt = torch.empty((b, 126), device=f64.device, dtype=torch.float32)
    L.run_regressor_quad(cv_s, cv_t.view((2, b) + tuple(cv_t.shape[1:])), L.get_quad(spatial, temporal),
                         [off_ref, off_tgt, tm_out[0], tm_out[1]])
    return off_ref, off_tgt


class OnlineStitcher:
    def __init__(self, nets, height, width, canvas=None, margin=0.03, warp_mode='NORMAL', fusion_mode='AVERAGE',
                 use_graph=True, grow='never', meshes_only=False, deterministic=False):
        """canvas: optional (wmin, wmax, hmin, hmax) in HR pixels (e.g. the offline bbox).
        deterministic: every push under the conv engine's geometry-only kernel policy (ops.deterministic): with the offline canvas the
        stream's frames equal the resident clip's (pipeline.run_two_view(..., deterministic=True)) bit for bit; ~1.6x slower pushes.
        meshes_only: no canvas, no render -- `push` returns the newly smoothed meshes (m1, m2) [k,7,9,2] (k = 7 on the 7th push,
        then 1) or None; the building block of ThreeViewOnlineStitcher, which captures the graph itself (use_graph is ignored).
        grow: 'never' -- the canvas fixed after the first window stays (frames whose mesh leaves it are cropped and COUNTED:
        `clipped_frames`); 'recapture' -- the canvas grows (and the steady-state graph is captured again) when a mesh comes within
        half the margin of its edge; `canvas_epoch` counts the growths, `hc` / `wc` / `bbox` change with them."""
        if grow not in ('never', 'recapture'):
            raise ValueError("grow must be 'never' or 'recapture'")
        self.grow = grow
        self.deterministic = bool(deterministic)
        self.meshes_only = bool(meshes_only)
        self.last_meshes = None
        self.canvas_epoch = 0
        self.watch_i = self.watch_f = None          # device-side overflow state of the current canvas (ops.canvas_watch)
        self._watch_totals = [0, 0, -1]             # frames seen / clipped / first clipped frame on EARLIER canvases
        self._host_watch = self._host_event = None  # grow='recapture': pinned copy of the state, one push behind
        self._near_handled = 0
        self.spatial, self.temporal, self.smooth = nets
        self.dev = next(self.spatial.parameters()).device
        self.h, self.w = height, width
        self.margin = margin
        self.warp_mode, self.fusion_mode = warp_mode, fusion_mode
        self.bbox = None if canvas is None else torch.tensor(canvas, dtype=torch.float32, device=self.dev)
        self.hc = self.wc = None
        self.nrigid = get_norm_mesh(get_rigid_mesh(1, height, width, device=self.dev), height, width).contiguous()
        self.frames_in = 0
        self.prev_feat = None            # TemporalNet stage-1 features of the previous frame, both views [2,45,60,128]
        self.prev_smotion = None         # [2,7,9,2]
        self.ring_smesh = [[], []]       # last WINDOW spatial meshes per view, each [1,7,9,2]
        self.ring_tsm = [[], []]
        self.ring_hr = []                # HR frames waiting for their smoothed mesh (only until the first window)
        self.use_graph = use_graph
        self.static = None               # steady-state buffers (inputs, rings, output) once the window is full
        self.graph = None
        self.graph_nodes = None          # nodes of the captured steady-state graph (None: not captured / not exposed)
        self.trunk_pair = None
        self.trunk_versions = None

    def _set_canvas(self):
        bb = self.bbox.cpu()
        self.hc = int((bb[3] - bb[2]).int())
        self.wc = int((bb[1] - bb[0]).int())
        self.watch_i, self.watch_f = ops.canvas_watch_state(1, self.dev)
        self._near_handled = 0
        self._host_event = None

    # ------------------------------------------------------------------ canvas overflow
    def _guard(self):
        """`near` threshold of the watcher in canvas-normalised units: half the margin the canvas was grown by."""
        return max(0.0, float(self.margin)) * 0.5 / (1.0 + 2.0 * max(0.0, float(self.margin))) * 2.0

    def overflow_report(self):
        """Synchronises.  -> {'frames_seen', 'clipped_frames', 'first_clipped_frame' (stream frame index, -1 = none), 'near_frames'
        (current canvas), 'canvas_epoch', 'needed_bbox' (wmin, wmax, hmin, hmax in HR px: the current canvas united with every
        mesh seen on it)}."""
        seen, clipped, first = self._watch_totals
        rep = {'frames_seen': seen, 'clipped_frames': clipped, 'first_clipped_frame': first, 'near_frames': 0,
               'canvas_epoch': self.canvas_epoch, 'needed_bbox': None}
        if self.watch_i is None:
            return rep
        wi = self.watch_i[0].cpu().tolist()
        if wi[1] > 0 and first < 0:
            rep['first_clipped_frame'] = seen + wi[2]
        rep['frames_seen'] = seen + wi[0]
        rep['clipped_frames'] = clipped + wi[1]
        rep['near_frames'] = wi[3]
        rep['needed_bbox'] = tuple(float(x) for x in self._needed_bbox(self.watch_f[0].cpu()))
        return rep

    @property
    def clipped_frames(self):
        """Frames emitted so far whose mesh reached outside the canvas they were rendered on (synchronises)."""
        return self.overflow_report()['clipped_frames']

    def _needed_bbox(self, wf):
        """Running normalised extents [xmin, xmax, ymin, ymax] on the current canvas -> union with the canvas, HR pixels."""
        bb = self.bbox.cpu()
        ow, oh = float(bb[1] - bb[0]), float(bb[3] - bb[2])
        x0 = float(bb[0]) + (min(float(wf[0]), -1.0) + 1.0) * ow / 2.0
        x1 = float(bb[0]) + (max(float(wf[1]), 1.0) + 1.0) * ow / 2.0
        y0 = float(bb[2]) + (min(float(wf[2]), -1.0) + 1.0) * oh / 2.0
        y1 = float(bb[2]) + (max(float(wf[3]), 1.0) + 1.0) * oh / 2.0
        return x0, x1, y0, y1

    def _regrow(self, wi, wf):
        """grow='recapture': re-fix the canvas around everything seen so far (+ margin), fresh watcher state, new output buffer,
        the steady-state graph captured again at the next push."""
        x0, x1, y0, y1 = self._needed_bbox(wf)
        gw, gh = self.margin * (x1 - x0), self.margin * (y1 - y0)
        old = [float(v) for v in self.bbox.cpu()]
        if max(abs(a - b) for a, b in zip(old, (x0 - gw, x1 + gw, y0 - gh, y1 + gh))) < 0.5:
            # nothing to grow by (e.g. margin 0 and a mesh that only touches the edge): keep the canvas, the output buffer and the
            # captured graph -- a recapture costs a sync, an allocation and a capture, and would be asked for again next push
            self._near_handled = int(wi[3])
            return False
        seen, clipped, first = self._watch_totals
        if wi[1] > 0 and first < 0:
            first = seen + wi[2]
        self._watch_totals = [seen + wi[0], clipped + wi[1], first]
        self.bbox = torch.tensor([x0 - gw, x1 + gw, y0 - gh, y1 + gh], dtype=torch.float32, device=self.dev)
        self._set_canvas()
        self.canvas_epoch += 1
        if self.static is not None:
            self.static['out'] = torch.empty((3, self.hc, self.wc), device=self.dev)
        self.graph = None
        return True

    def _poll_growth(self):
        """Start of a steady-state push (grow='recapture'): look at the watcher state of the push before, copied to pinned memory
        behind it -- no wait: if the copy has not landed yet the check happens one push later."""
        if self._host_event is None or not self._host_event.query():
            return
        if int(self._host_watch[0][0, 3]) > self._near_handled:
            # (rare path, synchronises: the device state may be a push ahead of the pinned copy -- take everything seen so far)
            self._regrow(self.watch_i[0].cpu().tolist(), self.watch_f[0].cpu())

    def _post_watch_copy(self):
        if self._host_watch is None:
            self._host_watch = (torch.empty((1, 4), dtype=torch.int32).pin_memory(), torch.empty((1, 4), dtype=torch.float32).pin_memory())
        self._host_watch[0].copy_(self.watch_i, non_blocking=True)
        self._host_watch[1].copy_(self.watch_f, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self._host_event = ev

    @torch.no_grad()
    def _render(self, hr1, hr2, mesh1, mesh2, out=None):
        """mesh* [1,7,9,2] LR-scale smoothed meshes of ONE frame -> stitched frame [3,Hc,Wc] (written to `out` if given)."""
        if FUSED_SPLINES:                    # control points + splines in one launch, the watcher inside the footprint launch
            src4, T4 = ops.stream_splines([mesh1, mesh2], 126, self.bbox, self.nrigid, self.h, self.w)
            return self._render_solved(hr1, hr2, src4[0], T4[0], out, watch=(self._guard(), self.watch_i, self.watch_f))
        src4 = ops.stream_normalize_watch([mesh1, mesh2], 126, self.bbox, self.h, self.w, self._guard(), self.watch_i,
                                          self.watch_f)                                      # [1,2,63,2], watcher updated
        src = src4[0]
        T = ops.tps_solve_shared(src, self.nrigid)
        return self._render_solved(hr1, hr2, src, T, out)

    def _direct(self):
        """Does the steady-state push launch its render itself (DIRECT_RENDER), outside the graph?"""
        return bool(DIRECT_RENDER and self.use_graph and self.fusion_mode == 'AVERAGE' and not self.meshes_only)

    def _render_solved(self, hr1, hr2, src, T, out=None, watch=None):
        """src [2,63,2] normalised control points on this stream's canvas, T [2,2,66] their splines -> stitched frame.
        watch = (guard, watch_i [1,4], watch_f [1,4]): the overflow watcher has not seen `src` yet."""
        if watch is not None and not (self.fusion_mode == 'AVERAGE' and pipeline.SKIP_OUTSIDE):
            ops.canvas_watch(src[None], watch[1], watch[2], watch[0])            # no footprint launch to carry it
            watch = None
        if self.fusion_mode == 'AVERAGE':
            fp = None
            if pipeline.SKIP_OUTSIDE:        # same footprint skipping as the offline render (pipeline.render_frames)
                fp = ops.render_footprints(src[None], T[None], self.h, self.w, self.hc, self.wc, watch=watch)[0]
            if out is _DEFER:
                self._deferred = (src, T, fp)
                return None
            return ops.render_average([hr1, hr2], src, T, self.hc, self.wc, self.warp_mode, out=out, footprint=fp)
        w = ops.tps_warp_views([hr1, hr2], src, T, self.hc, self.wc, self.warp_mode)
        res = ops.linear_blend(w[0, 0:3], w[1, 0:3], w[0, 3], w[1, 3])
        return res if out is None else out.copy_(res)

    # ------------------------------------------------------------------ steady state (window full, canvas fixed)
    # State tensors of the steady state (fixed addresses: the step is captured into a HIP graph):
    #   pair_s [2 views][prev, new][126]   spatial motions of the previous and the current pair
    #   pair_t [2 views][zero, new][126]   temporal motions (slot 0 stays zero: tsmotion of a 2-frame batch reads slot 1 only)
    #   ring   [smesh v0, smesh v1, tsm v0, tsm v1][7][126]     the sliding SmoothNet window
    #   prev_feat [2,45,60,128]            TemporalNet stage-1 features of the previous frame, both views
    _STATE = ('prev_feat', 'pair_s', 'ring')

    def _init_static(self):
        d = self.dev
        e = 126
        pair_s = torch.zeros((2, 2, e), device=d)
        pair_s[:, 0] = self.prev_smotion.reshape(2, e)
        ring = torch.stack([torch.cat(r, 0).reshape(WINDOW, e) for r in (self.ring_smesh[0], self.ring_smesh[1],
                                                                          self.ring_tsm[0], self.ring_tsm[1])], 0).contiguous()
        lr = torch.empty((2, 1, 3, pipeline.LR_H, pipeline.LR_W), device=d)       # both views back to back: one layout launch per push
        st = {'hr1': torch.empty((1, 3, self.h, self.w), device=d), 'hr2': torch.empty((1, 3, self.h, self.w), device=d),
              'lr1': lr[0], 'lr2': lr[1],
              'prev_feat': self.prev_feat.clone(), 'pair_s': pair_s, 'pair_t': torch.zeros((2, 2, e), device=d),
              'ring': ring, 'ts_out': torch.empty((2, 4, e), device=d),
              'out': None if self.meshes_only else torch.empty((3, self.hc, self.wc), device=d)}
        self.static = st

    def _step_static(self):
        """One steady-state push on the static buffers (capturable: no host sync, no data-dependent shapes; every result
        lands in place -- no torch op in the step besides the copy of the cached features)."""
        st = self.static
        f2, off1 = self._stage_a(st['lr1'], st['lr2'])
        self._stage_b(f2, off1, st['hr1'], st['hr2'], _DEFER if self._direct() else st['out'])

    def _stage_a(self, lr1, lr2):
        """First half of a steady-state push -- it touches NO stream state: both nets' stage-1 trunks on the two LR frames (one
        grouped launch per layer: SpatialNet and TemporalNet read the same frames through trunks of identical architecture), then
        SpatialNet's stage-2 trunk, contextual correlation and regressNet1 -> (f2 [2(net),2(view),45,60,128], offset_1 [1,8])."""
        if self.trunk_pair is None:
            self.trunk_pair = L.pair_trunks(self.spatial._prepared()['s1'], self.temporal._prepared()['s1'])
            self.trunk_versions = self._versions()
        f2 = L.run_stage1_pair([lr1, lr2], self.trunk_pair)
        if not L.QUAD:
            return f2, None
        return f2, _heads_a(self.spatial, f2[0], 1)

    def _stage_b(self, f2, off1, hr1, hr2, out):
        """Second half: everything that reads or advances the stream's state (cached features, previous motions, the sliding
        window) and the render of the new frame into `out`."""
        st = self.static
        e = 126
        ps, pt = st['pair_s'], st['pair_t']
        feat = f2[1]
        tm_out = (pt[0, 1:2], pt[1, 1:2])
        if off1 is None:
            off1, off_ref, off_tgt = _spatial_temporal_heads(self.spatial, self.temporal, f2[0], st['prev_feat'], feat, 1, tm_out)
        else:
            off_ref, off_tgt = _heads_b(self.spatial, self.temporal, f2[0], off1, st['prev_feat'], feat, 1, tm_out)
        ops.spatial_meshes(off1, off_ref, off_tgt, pipeline.LR_H, pipeline.LR_W,
                           out=(ps[0, 1].view(1, 7, 9, 2), ps[1, 1].view(1, 7, 9, 2)))
        st['prev_feat'].copy_(feat)
        # tsmotion of both views as ONE batch of 4 frames (v0 prev, v0 new, v1 prev, v1 new): frame k pairs with frame k - 1,
        # rows 1 and 3 are this pair's; row 2 (view 1's previous frame against view 0's new one) is computed and ignored
        ops.tsmotion(ps.view(4, 7, 9, 2), pt.view(4, 7, 9, 2), pipeline.LR_H, pipeline.LR_W, out=(st['ts_out'][0], st['ts_out'][1]))
        # shift the four rings by one frame, append this pair's rows, and make the current spatial motions the previous ones
        ops.window_push(st['ring'], st['ts_out'], [1 * e, 3 * e, 5 * e, 7 * e], state=ps, blocks=2, block=e, stride=2 * e, delta=e)
        r = st['ring'].view(4, WINDOW, 7, 9, 2)
        outs, _ = self.smooth.run_windows(r[0], r[1], r[2], r[3], 1, WINDOW, 1, 1)
        m1, m2 = outs['smooth_mesh1'][0], outs['smooth_mesh2'][0]
        if self.meshes_only:
            self.last_meshes = (m1[-1:], m2[-1:])
            return
        self._render(hr1, hr2, m1[-1:], m2[-1:], out=out)

    def _versions(self):
        return (self.spatial.weights_version, self.temporal.weights_version, self.smooth.weights_version)

    def _push_static(self, hr1, hr2, lr1, lr2, u8=None):
        st = self.static
        if self.grow == 'recapture' and not self.meshes_only:
            self._poll_growth()
        if self.trunk_pair is not None and self.trunk_versions != self._versions():
            # a net was reloaded / moved since the twin trunk was stacked and the graph captured: both hold the OLD
            # weights (the graph by address); rebuild and recapture instead of silently stitching with stale filters
            self.trunk_pair = None
            self.graph = None
        direct = self._direct()
        if u8 is not None:                   # decoded uint8 frames (push_u8): the cv2-exact resize writes the graph's LR buffers itself
            ops.ingest_u8(u8[0][None], pipeline.LR_H, pipeline.LR_W, want_hr=False, lr_out=st['lr1'])
            ops.ingest_u8(u8[1][None], pipeline.LR_H, pipeline.LR_W, want_hr=False, lr_out=st['lr2'])
        else:
            if not self.meshes_only and not direct:  # (meshes_only: the frames are not looked at, None will do; direct: rendered in place)
                st['hr1'].copy_(hr1.reshape(st['hr1'].shape)); st['hr2'].copy_(hr2.reshape(st['hr2'].shape))
            st['lr1'].copy_(lr1.reshape(st['lr1'].shape)); st['lr2'].copy_(lr2.reshape(st['lr2'].shape))
        if not self.use_graph:
            self._step_static()
        elif self.graph is None:
            # capture: the eager warm-up runs on a copy of the state so that this push is applied exactly once
            keep = {k: v.clone() for k, v in st.items() if k in self._STATE}
            # (the watcher counts every run of the step; meshes_only has no canvas and no watcher)
            keep_w = None if self.meshes_only else (self.watch_i.clone(), self.watch_f.clone())
            side = _warmup_stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._step_static()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            for k, v in keep.items():
                st[k].copy_(v)
            if keep_w is not None:
                self.watch_i.copy_(keep_w[0]); self.watch_f.copy_(keep_w[1])
            g = _new_graph()
            with torch.cuda.graph(g):
                self._step_static()
            self.graph = g
            self.graph_nodes = _graph_nodes(g)
            for k, v in keep.items():          # capture does not execute: state is still the pre-push state
                st[k].copy_(v)
            self.graph.replay()
        else:
            self.graph.replay()
        self.frames_in += 1
        if self.meshes_only:             # the newest smoothed meshes (m1, m2) [1,7,9,2]: copies, the step's own tensors are reused
            return tuple(m.clone() for m in self.last_meshes)
        if self.grow == 'recapture':
            self._post_watch_copy()
        if direct:                       # the graph left splines and footprints; the render reads the caller's frames, writes a new tensor
            src, T, fp = self._deferred
            if u8 is not None:               # uint8 frames in, the uint8 video frame out: no fp32 frame planes, no fp32 canvas
                return [ops.render_average_u8([u8[0], u8[1]], src, T, self.hc, self.wc, self.warp_mode, footprint=fp)]
            return [ops.render_average([hr1.reshape(st['hr1'].shape), hr2.reshape(st['hr2'].shape)], src, T, self.hc, self.wc,
                                       self.warp_mode, footprint=fp)]
        return [st['out'].clone()]

    @torch.no_grad()
    def push(self, hr1, hr2, lr1, lr2):
        """One frame pair: hr* [1,3,H,W] (0..255), lr* [1,3,360,480] ([-1,1]), device tensors.
        -> list of newly stitched frames (empty for the first 6 pushes, 7 frames on the 7th, then one per push).
        meshes_only: -> None for the first 6 pushes, then (m1, m2) [k,7,9,2] (k = 7 on the 7th push, then 1)."""
        with ops.deterministic(self.deterministic):
            return self._push(hr1, hr2, lr1, lr2)

    @torch.no_grad()
    def push_u8(self, img1, img2):
        """One DECODED frame pair: img* uint8 [H,W,3] device tensors in cv2.imread's layout and channel order (the reference's frame
        loop, test_online_tra.py:252-278) -> list of stitched VIDEO frames uint8 [Hc,Wc,3] (`.astype(np.uint8)` of the fused values,
        :413), empty for the first 6 pushes, 7 frames on the 7th, then one per push.  Byte for byte ops.ingest_u8 -> push ->
        ops.canvas_to_u8; in the steady state (DIRECT_RENDER, fusion AVERAGE) the cv2-exact resize writes the graph's LR inputs and
        the render samples the uint8 frames and writes the uint8 frame itself: no fp32 frame planes, no fp32 canvas."""
        if self.meshes_only:
            raise ValueError('push_u8 renders frames: not for meshes_only stitchers')
        if img1.dtype != torch.uint8 or img1.dim() != 3 or img1.shape[-1] != 3 or tuple(img1.shape) != tuple(img2.shape):
            raise ValueError('push_u8 takes two uint8 [H,W,3] frames')
        with ops.deterministic(self.deterministic):
            if self.static is not None and self._direct() and type(self)._push_static is OnlineStitcher._push_static:
                return self._push_static(None, None, None, None, u8=(img1.contiguous(), img2.contiguous()))
            hr, lr = ops.ingest_u8(torch.stack((img1, img2), 0), pipeline.LR_H, pipeline.LR_W)
            frames = self._push(hr[0:1], hr[1:2], lr[0:1], lr[1:2])
            return [ops.canvas_to_u8(f.reshape((1,) + tuple(f.shape[-3:])))[0] for f in frames]

    def _push(self, hr1, hr2, lr1, lr2):
        if self.static is not None:
            return self._push_static(hr1, hr2, lr1, lr2)
        t = self.frames_in
        # spatial warp of this pair
        o = build_SpatialNet(self.spatial, lr1, lr2)
        smotion = torch.cat((o['motion1'], o['motion2']), 0)                        # [2,7,9,2]
        # temporal warp: only the new frame goes through the trunk, the previous features are cached
        feat = self.temporal.features([lr1, lr2])                                   # [2,45,60,128]
        if t == 0:
            tmotion = torch.zeros_like(smotion)
        else:
            tmotion = self.temporal.motions_from_features(self.prev_feat, feat)
        self.prev_feat = feat
        # tsmotion of frame t from smotion_{t-1} (frame 0: zero)
        for v in range(2):
            if t == 0:
                smesh = get_rigid_mesh(1, pipeline.LR_H, pipeline.LR_W, device=self.dev) + smotion[v:v + 1]
                tsm = torch.zeros_like(smesh)
            else:
                pair_s = torch.cat((self.prev_smotion[v:v + 1], smotion[v:v + 1]), 0)
                pair_t = torch.cat((torch.zeros_like(tmotion[v:v + 1]), tmotion[v:v + 1]), 0)
                sm2, ts2 = ops.tsmotion(pair_s, pair_t, pipeline.LR_H, pipeline.LR_W)
                smesh, tsm = sm2[1:2], ts2[1:2]
            self.ring_smesh[v] = (self.ring_smesh[v] + [smesh])[-WINDOW:]
            self.ring_tsm[v] = (self.ring_tsm[v] + [tsm])[-WINDOW:]
        self.prev_smotion = smotion
        self.frames_in += 1
        if self.hc is None and not self.meshes_only:
            self.ring_hr.append((hr1, hr2))
        if self.frames_in < WINDOW:
            return None if self.meshes_only else []
        # smooth the current window (first tsmotion of the window counts as zero)
        sm = [torch.cat(self.ring_smesh[v], 0).contiguous() for v in range(2)]
        ts = [torch.cat(self.ring_tsm[v], 0).contiguous() for v in range(2)]
        outs, _ = self.smooth.run_windows(sm[0], sm[1], ts[0], ts[1], 1, WINDOW, 1, 1)
        m1, m2 = outs['smooth_mesh1'][0], outs['smooth_mesh2'][0]                   # [7,7,9,2]
        if self.meshes_only:                                                       # first window: 7 meshes, then the static step
            self.ring_hr = []
            self._init_static()
            self.last_meshes = (m1, m2)
            return self.last_meshes
        if self.hc is None:                                                        # first window: fix the canvas, emit 7
            if self.bbox is None:
                bb = ops.mesh_bbox([m1, m2], self.h, self.w).cpu()
                gw, gh = self.margin * (bb[1] - bb[0]), self.margin * (bb[3] - bb[2])
                self.bbox = torch.stack((bb[0] - gw, bb[1] + gw, bb[2] - gh, bb[3] + gh)).to(self.dev)
            self._set_canvas()
            frames = [self._render(h1, h2, m1[i:i + 1], m2[i:i + 1]) for i, (h1, h2) in enumerate(self.ring_hr)]
            self.ring_hr = []
            self._init_static()              # from the next push on: static buffers (+ HIP graph)
            return frames
        return [self._render(hr1, hr2, m1[-1:], m2[-1:])]


# Steady-state AVERAGE render OUTSIDE the captured graph, on the caller's own HR frames and into a fresh tensor: the graph ends with the
# splines / footprints, the push saves the copies of the HR frames into static buffers and the clone of the static canvas (720p, three
# views: 66 + 50 MB of HBM traffic, ~35 us of a 1.1 ms push).  Same kernel, same operands: frames bit-identical.
DIRECT_RENDER = os.environ.get('SS_DIRECT_RENDER', '1') != '0'
_DEFER = object()                      # `out=_DEFER`: compute splines and footprints, leave the render launch to the push
FUSED_SPLINES = os.environ.get('SS_FUSED_SPLINES', '1') != '0'   # ThreeViewOnlineStitcher: composition + splines in one launch
PIPE_STREAM_CANDIDATES = 5        # streams tried pairwise by _TwoInFlight._pick_streams
PIPE_PROBE_PUSHES = 8


class _TwoInFlight:
    """Mixin: the steady-state push as two halves on two HIP streams, two pushes in flight (PipelinedOnlineStitcher,
    PipelinedMultiOnlineStitcher).  A batch-1 push is a dependent chain of ~80 small launches that leaves most of the chip idle
    (profiles/r06_stream_timeline.txt: 0.86 ms; a graph node costs >= 4.5 us whatever it does), and nothing inside ONE push overlaps
    any further (LAB_NOTES R6.1-2).  Consecutive pushes do: the first half of a push -- both trunks, SpatialNet's stage-2 trunk,
    contextual correlation, regressNet1 (`_stage_a`) -- touches no stream state, so push t + 1's first half runs beside push t's second
    half (`_stage_b`: decomposition, cost volumes, regressor heads, tsmotion, sliding window, SmoothNet, render).  Each half is its own
    HIP graph per buffer parity; hand-over buffers are double-buffered, events order the halves.  Per frame the launches and their
    operands are the plain stitcher's: results are bit-identical.  What changes is WHEN a result is handed out: `push` returns the
    frames of the PREVIOUS push (valid on the caller's stream), `flush()` the last ones.
    A subclass provides _pipe_alloc / _pipe_load / _run_a / _run_b / _pipe_take / _pipe_state / _pipe_empty."""

    def _pipe_init(self):
        self.pipe = None
        self._pending = None             # (event behind it, result) of the newest enqueued push
        self._t = 0

    def _capture_pipe(self):
        """Warm both halves up eagerly on a copy of the state (parity 0 holds the current push's inputs), then capture each half for
        each buffer parity.  Captures do not execute: the state is the pre-push state afterwards."""
        torch.cuda.synchronize(self.dev)
        state = self._pipe_state()
        keep = [t.clone() for t in state]
        side = _warmup_stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self._run_a(0)
            self._run_b(0)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        for t, v in zip(state, keep):
            t.copy_(v)
        nodes = 0
        for p in (0, 1):
            for key, fn in (('ga', self._run_a), ('gb', self._run_b)):
                g = _new_graph()
                with torch.cuda.graph(g):
                    fn(p)
                self.pipe[key][p] = g
                if key == 'gb':              # (direct render: the second half left this parity's splines and footprints)
                    self.pipe.setdefault('deferred', [None, None])[p] = self._pipe_deferred()
                if p == 0:
                    n = _graph_nodes(g)
                    nodes = None if nodes is None or n is None else nodes + n
        self.graph_nodes = nodes             # nodes of one push (both halves)
        for t, v in zip(state, keep):
            t.copy_(v)
        torch.cuda.synchronize(self.dev)

    def _enqueue(self, p, sa, sb, load=None):
        """One push's two halves behind the events that order them; -> the event behind the second half."""
        P = self.pipe
        cur = torch.cuda.current_stream(self.dev)
        ev_in = torch.cuda.Event()
        ev_in.record(cur)
        sa.wait_event(ev_in)
        sb.wait_event(ev_in)
        if P['eB'][p] is not None:
            sa.wait_event(P['eB'][p])                # the second half of push t - 2 has read this parity's hand-over buffers
        if load is not None:
            load()
        with torch.cuda.stream(sa):
            P['ga'][p].replay()
            ea = torch.cuda.Event()
            ea.record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(ea)
            P['gb'][p].replay()
            eb = torch.cuda.Event()
            eb.record(sb)
        P['eB'][p] = eb
        return eb

    def _pick_streams(self):
        """Choose the two streams by measurement.  Which hardware queue (and which of the command processor's pipes) a HIP stream
        lands on is the runtime's choice -- round-robin over GPU_MAX_HW_QUEUES by creation order, so it depends on every stream the
        process has made before -- and two streams that share a queue, or whose queues the firmware serves from one pipe, turn the
        0.89 ms three-view push into 1.3 - 4 ms (LAB_NOTES R6.5; priorities do not help).  The graphs replay on any stream, so: make
        PIPE_STREAM_CANDIDATES streams, replay the captured halves with the push's own event pattern on every pair, keep the fastest
        pair, restore the state the replays advanced.  ~0.1 s, once."""
        state = self._pipe_state()
        keep = [t.clone() for t in state]
        cands = [torch.cuda.Stream(self.dev) for _ in range(PIPE_STREAM_CANDIDATES)]
        cur = torch.cuda.current_stream(self.dev)
        best, self.stream_probe_ms = None, []
        for i, sa in enumerate(cands):
            for sb in cands[i + 1:]:
                ms = None
                for pushes in (2, PIPE_PROBE_PUSHES):        # the first pass pages the graphs onto the streams
                    self.pipe['eB'] = [None, None]
                    torch.cuda.synchronize(self.dev)
                    t0 = time.perf_counter()
                    prev = None
                    for t in range(pushes):
                        eb = self._enqueue(t & 1, sa, sb)
                        if prev is not None:
                            cur.wait_event(prev)
                        prev = eb
                    torch.cuda.synchronize(self.dev)
                    ms = (time.perf_counter() - t0) / pushes * 1e3
                self.stream_probe_ms.append(round(ms, 4))
                if best is None or ms < best[0]:
                    best = (ms, sa, sb)
        self.pipe['eB'] = [None, None]
        for t, v in zip(state, keep):
            t.copy_(v)
        torch.cuda.synchronize(self.dev)
        self.pipe['sa'], self.pipe['sb'] = best[1], best[2]

    def _push_pipelined(self, *inputs):
        if self.pipe is None:
            self.pipe = dict(self._pipe_alloc(), sa=None, sb=None, ga=[None, None], gb=[None, None], eB=[None, None])
        if self.trunk_pair is not None and self.trunk_versions != self._versions():
            # a net was reloaded / moved: the twin trunk and ALL four graphs hold the old weights by address -- drain, recapture
            torch.cuda.synchronize(self.dev)
            self.trunk_pair = None
            self.pipe['ga'] = [None, None]
            self.pipe['gb'] = [None, None]
        P, p = self.pipe, self._t & 1
        if P['ga'][p] is None:
            # the capture warm-up reads parity 0's inputs: load them on the caller's stream first
            cur = torch.cuda.current_stream(self.dev)
            self._pipe_load(p, cur, cur, *inputs)
            self._capture_pipe()
            if P['sa'] is None:
                self._pick_streams()
        sa, sb = P['sa'], P['sb']
        eb = self._enqueue(p, sa, sb, lambda: self._pipe_load(p, sa, sb, *inputs))
        with torch.cuda.stream(sb):
            result = self._pipe_take(p, *inputs)
            eb = torch.cuda.Event()
            eb.record(sb)
        P['eB'][p] = eb
        prev, self._pending = self._pending, (eb, result)
        self._t += 1
        self.frames_in += 1
        return self._hand_out(prev)

    def _pipe_deferred(self):
        return getattr(self, '_deferred', None)

    def _hand_out(self, item):
        if item is None:
            return self._pipe_empty()
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(item[0])
        for t in _tensors_of(item[1]):
            t.record_stream(cur)
        return item[1]

    def flush(self):
        """-> the result of the newest push (what `push` would have returned one push later), valid on the caller's stream."""
        item, self._pending = self._pending, None
        return self._hand_out(item)

    def flush_u8(self):
        """`flush()` for streams fed through push_u8: the last result as uint8 video frames."""
        def conv(x):
            if torch.is_tensor(x):
                return x if x.dtype == torch.uint8 else ops.canvas_to_u8(x.reshape((1,) + tuple(x.shape[-3:])))[0]
            return [conv(y) for y in x]
        return conv(self.flush())

    def overflow_report(self):
        if getattr(self, 'pipe', None) is not None:
            torch.cuda.synchronize(self.dev)
        return super().overflow_report()


def _tensors_of(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (list, tuple)):
        for y in x:
            yield from _tensors_of(y)


class PipelinedOnlineStitcher(_TwoInFlight, OnlineStitcher):
    """OnlineStitcher with TWO pushes in flight (round 6; opt-in; see _TwoInFlight).  Frames bit-identical to OnlineStitcher
    (tests/test_gpu_round6.py), handed out one push late:
        st = PipelinedOnlineStitcher(nets, H, W)
        for pair in stream: for frame in st.push(*pair): ...
        for frame in st.flush(): ...
    0.64 ms per push = 1559 frames/s at 720p against 0.85 ms / 1171.  The canvas is fixed after the first window (grow='never';
    overflow is counted as in OnlineStitcher)."""

    def __init__(self, nets, height, width, canvas=None, margin=0.03, warp_mode='NORMAL', fusion_mode='AVERAGE', deterministic=False):
        OnlineStitcher.__init__(self, nets, height, width, canvas, margin, warp_mode, fusion_mode, use_graph=True, grow='never',
                                meshes_only=False, deterministic=deterministic)
        if not L.QUAD:
            raise ValueError('PipelinedOnlineStitcher needs the shared regressor launches (SS_QUAD_REGRESSOR=1)')
        self._pipe_init()

    def _pipe_alloc(self):
        d = self.dev
        two = lambda *shape: [torch.empty(shape, device=d) for _ in range(2)]
        hr = (lambda: [None, None]) if self._direct() else (lambda: two(1, 3, self.h, self.w))     # direct render: no copies of the frames
        return {'lr': two(2, 1, 3, pipeline.LR_H, pipeline.LR_W), 'hr1': hr(), 'hr2': hr(),
                'f2': two(2, 2, pipeline.LR_H // 8, pipeline.LR_W // 8, 128), 'off1': two(1, 8),
                'out': [None, None] if self._direct() else two(3, self.hc, self.wc)}

    def _pipe_state(self):
        return [self.static[k] for k in self._STATE] + [self.watch_i, self.watch_f]

    def _pipe_empty(self):
        return []

    def _pipe_load(self, p, sa, sb, hr1, hr2, lr1, lr2):
        P = self.pipe
        if hr1.dtype == torch.uint8:         # push_u8: decoded [H,W,3] frames -- the cv2-exact resize writes this parity's LR inputs
            with torch.cuda.stream(sa):
                ops.ingest_u8(hr1[None], pipeline.LR_H, pipeline.LR_W, want_hr=False, lr_out=P['lr'][p][0])
                ops.ingest_u8(hr2[None], pipeline.LR_H, pipeline.LR_W, want_hr=False, lr_out=P['lr'][p][1])
            for t in (hr1, hr2):
                t.record_stream(sa)
                t.record_stream(sb)
            return
        with torch.cuda.stream(sa):
            P['lr'][p][0].copy_(lr1.reshape(P['lr'][p][0].shape))
            P['lr'][p][1].copy_(lr2.reshape(P['lr'][p][1].shape))
        if not self._direct():               # (direct render: the second stream's render reads the caller's frames)
            with torch.cuda.stream(sb):
                P['hr1'][p].copy_(hr1.reshape(P['hr1'][p].shape))
                P['hr2'][p].copy_(hr2.reshape(P['hr2'][p].shape))
        for t in (hr1, hr2):
            t.record_stream(sb)
        for t in (lr1, lr2):
            t.record_stream(sa)

    def _run_a(self, p):
        P = self.pipe
        f2, off1 = self._stage_a(P['lr'][p][0], P['lr'][p][1])
        P['f2'][p].copy_(f2)
        P['off1'][p].copy_(off1)

    def _run_b(self, p):
        P = self.pipe
        self._stage_b(P['f2'][p], P['off1'][p], P['hr1'][p], P['hr2'][p], _DEFER if self._direct() else P['out'][p])

    def _pipe_take(self, p, hr1, hr2, lr1, lr2):
        P = self.pipe
        if self._direct():
            src, T, fp = P['deferred'][p]
            if hr1.dtype == torch.uint8:     # push_u8: the uint8 video frame straight from the uint8 frames
                return [ops.render_average_u8([hr1, hr2], src, T, self.hc, self.wc, self.warp_mode, footprint=fp)]
            shp = (1, 3, self.h, self.w)
            return [ops.render_average([hr1.reshape(shp), hr2.reshape(shp)], src, T, self.hc, self.wc,
                                       self.warp_mode, footprint=fp)]
        return [P['out'][p].clone()]

    def _push_static(self, hr1, hr2, lr1, lr2):
        return self._push_pipelined(hr1, hr2, lr1, lr2)

    @torch.no_grad()
    def push_u8(self, img1, img2):
        """OnlineStitcher.push_u8 with two pushes in flight: the frames of the PREVIOUS push come back (uint8 [Hc,Wc,3]); `flush_u8()`
        for the last.  A stream is fed through push_u8 or through push, not both."""
        if self.static is not None and self._direct():
            if img1.dtype != torch.uint8 or img1.dim() != 3 or img1.shape[-1] != 3 or tuple(img1.shape) != tuple(img2.shape):
                raise ValueError('push_u8 takes two uint8 [H,W,3] frames')
            with ops.deterministic(self.deterministic):
                return self._push_pipelined(img1.contiguous(), img2.contiguous(), None, None)
        return OnlineStitcher.push_u8(self, img1, img2)


class MultiOnlineStitcher:
    """S independent live video pairs advancing one frame per push as ONE batch (the sharding unit of the
    path -- independent streams -- inside one GPU).  A single stream at batch 1 is launch-bound (~150 small launches per
    pushed pair); S streams run the same launches on S-fold batches, so the aggregate frame rate grows almost linearly until
    the kernels fill the chip.  Every stream keeps its own sliding window, cached TemporalNet features and FIXED canvas
    (canvases may differ in size); per stream and push the arithmetic is the single-stream stitcher's (test_online_tra.py:
    284-392 with k = t) -- the networks see the S pairs as one batch, the render goes stream by stream.

        st = MultiOnlineStitcher(nets, 720, 1280, streams=8)
        frames = st.push(hr1, hr2, lr1, lr2)      # hr* [S,3,H,W], lr* [S,3,360,480] -> S lists of new frames [3,Hc_s,Wc_s]

    The first WINDOW pushes fill every stream's window through S single-stream stitchers (eager); from then on the step runs
    on static buffers with a leading S and is captured into one HIP graph.  Results equal S single-stream stitchers up to the
    conv engine's launch-size-dependent kernel choice (Winograd / implicit GEMM / split-K are picked per launch size, so a
    batch of S sums in another order than S batches of one: motions differ by ~1e-5 px); within a batch every stream's result
    is independent of its neighbours bit for bit (tests/test_gpu_round4.py)."""

    def __init__(self, nets, height, width, streams, canvases=None, margin=0.03, warp_mode='NORMAL', fusion_mode='AVERAGE',
                 use_graph=True, grow='never', meshes_only=False, deterministic=False, chain=False):
        """grow: as OnlineStitcher -- 'never' counts the frames whose mesh left their stream's canvas (`clipped_frames`, per
        stream), 'recapture' re-fixes the canvases of the streams that come near an edge and captures the graph again.
        meshes_only: no canvases, no render -- `push` returns the S streams' newly smoothed meshes (m1, m2) [S,k,7,9,2] (k = 7 on the
        7th push, then 1) or None; ThreeViewOnlineStitcher runs its two pair chains as such a batch of two and captures the graph
        itself (use_graph is ignored)."""
        if grow not in ('never', 'recapture'):
            raise ValueError("grow must be 'never' or 'recapture'")
        self.grow = grow
        self.deterministic = bool(deterministic)      # geometry-only kernel policy: S batched streams == S single streams, bit for bit
        # chain: the S pairs are the neighbours of a chain of S + 1 views (pair s = views s, s + 1: ThreeViewOnlineStitcher's two
        # chains).  The steady-state step then takes the S + 1 LR frames once (static['lrc'] [S+1,3,360,480]) and every inner view
        # passes the trunks ONCE; the window-fill pushes go pair by pair as usual.  meshes_only streams only.
        self.chain = bool(chain)
        if self.chain and not meshes_only:
            raise ValueError('chain=True needs meshes_only=True')
        self.meshes_only = bool(meshes_only)
        self.last_meshes = None
        self._host_watch = self._host_event = None
        self.nets = nets
        self.spatial, self.temporal, self.smooth = nets
        self.dev = next(self.spatial.parameters()).device
        self.h, self.w, self.S = height, width, int(streams)
        if self.S < 1:
            raise ValueError('streams must be >= 1')
        if canvases is not None and len(canvases) != self.S:
            raise ValueError('one canvas per stream')
        self.warp_mode, self.fusion_mode = warp_mode, fusion_mode
        self.single = [OnlineStitcher(nets, height, width, None if canvases is None else canvases[s], margin, warp_mode,
                                      fusion_mode, use_graph=False, meshes_only=meshes_only, deterministic=deterministic)
                       for s in range(self.S)]                   # (growth is handled here, batched)
        self.use_graph = use_graph
        self.static = None
        self.graph = None
        self.graph_nodes = None
        self.trunk_pair = None
        self.trunk_versions = None
        self.frames_in = 0

    @property
    def canvas_sizes(self):
        """[(Hc, Wc)] per stream (None before the first window is complete)."""
        return [(s.hc, s.wc) for s in self.single]

    def _versions(self):
        return (self.spatial.weights_version, self.temporal.weights_version, self.smooth.weights_version)

    _STATE = ('prev_feat', 'pair_s', 'ring')

    def _init_static(self):
        """Batched steady-state buffers from the S single-stream states (each has just completed its first window):
          pair_s / pair_t [2 views][prev, new][S][126], ring [4 kinds][S][7][126], prev_feat [2 views * S,45,60,128] view-major
          (chain: [S + 1,45,60,128])."""
        d, S, e = self.dev, self.S, 126
        one = [s.static for s in self.single]
        lr = torch.empty((2, S, 3, pipeline.LR_H, pipeline.LR_W), device=d)       # both views back to back: one layout launch per push
        hr = None if self.meshes_only else torch.empty((2, S, 3, self.h, self.w), device=d)
        st = {'hr1': None if hr is None else hr[0], 'hr2': None if hr is None else hr[1],
              'lr1': lr[0], 'lr2': lr[1],
              'lrc': torch.empty((S + 1, 3, pipeline.LR_H, pipeline.LR_W), device=d) if self.chain else None,
              # (chain: the S + 1 views once -- pair s holds views s and s + 1; ops.cost_volume(chain=S) pairs them up)
              'prev_feat': (torch.stack([o['prev_feat'][0] for o in one] + [one[-1]['prev_feat'][1]], 0).contiguous() if self.chain else
                            torch.cat([torch.stack([o['prev_feat'][v] for o in one], 0) for v in range(2)], 0).contiguous()),
              'pair_s': torch.stack([o['pair_s'] for o in one], 2).contiguous(),           # [2,2,S,126]
              'pair_t': torch.zeros((2, 2, S, e), device=d),
              'ring': torch.stack([o['ring'] for o in one], 1).contiguous(),               # [4,S,7,126]
              'ts_out': torch.empty((2, 4 * S, e), device=d),
              'out': None, 'out_all': None}
        self.static = st
        if not self.meshes_only:
            st['bboxes'] = torch.stack([s.bbox for s in self.single], 0).contiguous()        # [S,4] the streams' fixed canvases
            st['watch_i'] = torch.cat([s.watch_i for s in self.single], 0).contiguous()       # [S,4] overflow state (ops.canvas_watch)
            st['watch_f'] = torch.cat([s.watch_f for s in self.single], 0).contiguous()
            self._alloc_outputs()
        for s in self.single:                     # the per-stream buffers are not needed any more (bbox / canvas stay)
            s.static = None

    def _alloc_outputs(self):
        st, S, d = self.static, self.S, self.dev
        st['out_all'] = None
        if len({(s.hc, s.wc) for s in self.single}) == 1:       # equal canvas sizes: one render launch for all streams
            st['out_all'] = torch.empty((S, 3, self.single[0].hc, self.single[0].wc), device=d)
            st['out'] = [st['out_all'][s] for s in range(S)]
        else:
            st['out'] = [torch.empty((3, s.hc, s.wc), device=d) for s in self.single]

    # ------------------------------------------------------------------ canvas overflow (per stream)
    def overflow_report(self):
        """Synchronises.  -> one dict per stream (OnlineStitcher.overflow_report)."""
        if self.static is None or self.meshes_only:          # (meshes_only: no canvases, nothing to overflow -- the empty reports)
            return [one.overflow_report() for one in self.single]
        wi, wf = self.static['watch_i'].cpu(), self.static['watch_f'].cpu()
        reps = []
        for s, one in enumerate(self.single):
            seen, clipped, first = one._watch_totals
            w = wi[s].tolist()
            if w[1] > 0 and first < 0:
                first = seen + w[2]
            reps.append({'frames_seen': seen + w[0], 'clipped_frames': clipped + w[1], 'first_clipped_frame': first,
                         'near_frames': w[3], 'canvas_epoch': one.canvas_epoch,
                         'needed_bbox': tuple(float(x) for x in one._needed_bbox(wf[s]))})
        return reps

    @property
    def clipped_frames(self):
        """Per stream: frames emitted so far whose mesh reached outside the canvas they were rendered on (synchronises)."""
        return [r['clipped_frames'] for r in self.overflow_report()]

    def _poll_growth(self):
        if self._host_event is None or not self._host_event.query():
            return
        near = self._host_watch[0][:, 3].tolist()
        cand = [s for s, one in enumerate(self.single) if near[s] > one._near_handled]
        if not cand:
            return
        st = self.static
        wi, wf = st['watch_i'].cpu(), st['watch_f'].cpu()              # (rare path: synchronises)
        fresh_i, fresh_f = ops.canvas_watch_state(1, self.dev)
        grown = False
        for s in cand:
            one = self.single[s]
            if not one._regrow(wi[s].tolist(), wf[s]):                 # nothing to grow by: canvas, outputs and graph stay
                continue
            grown = True
            st['bboxes'][s].copy_(one.bbox)
            st['watch_i'][s].copy_(fresh_i[0]); st['watch_f'][s].copy_(fresh_f[0])
        if grown:
            self._alloc_outputs()
            self.graph = None
        self._host_event = None

    def _post_watch_copy(self):
        st = self.static
        if self._host_watch is None:
            self._host_watch = (torch.empty((self.S, 4), dtype=torch.int32).pin_memory(), torch.empty((self.S, 4), dtype=torch.float32).pin_memory())
        self._host_watch[0].copy_(st['watch_i'], non_blocking=True)
        self._host_watch[1].copy_(st['watch_f'], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self._host_event = ev

    def _step_static(self):
        st = self.static
        f64, feat, off1 = self._stage_a(st['lrc'] if self.chain else [st['lr1'], st['lr2']])
        self._stage_b(f64, feat, off1, st['hr1'], st['hr2'], st['out_all'], st['out'], defer=self._direct())

    def _direct(self):
        """DIRECT_RENDER: the captured step stops at splines + footprints, the push renders from the caller's frames."""
        return bool(DIRECT_RENDER and self.use_graph and self.fusion_mode == 'AVERAGE' and not self.meshes_only)

    def _stage_a(self, lr):
        """First half of a steady-state push (no stream state): trunks, SpatialNet's stage-2 trunk, CCL, regressNet1.
        lr: [lr1, lr2] ([S,3,360,480] each) or, chain mode, the S + 1 views' frames [S+1,3,360,480] (each view once)
        -> (f64 SpatialNet trunk features, feat TemporalNet features [2S, view-major] (chain: [S + 1], each view once), offset_1 [S,8] | None)."""
        S = self.S
        if self.trunk_pair is None:
            self.trunk_pair = L.pair_trunks(self.spatial._prepared()['s1'], self.temporal._prepared()['s1'])
            self.trunk_versions = self._versions()
        if self.chain:
            fc = L.run_stage1_pair([lr], self.trunk_pair)                          # [2(net), S + 1 views, 45,60,128]: each view once
            feat = fc[1]                                                           # TemporalNet's features of the S + 1 views, each once
            f64 = fc[0]
        else:
            f2 = L.run_stage1_pair(list(lr), self.trunk_pair)                      # [2(net), 2S (view-major), 45,60,128]
            feat, f64 = f2[1], f2[0]
        return f64, feat, (_heads_a(self.spatial, f64, S, self.chain) if L.QUAD else None)

    def _stage_b(self, f64, feat, off1, hr1, hr2, out_all, out, defer=False):
        """Second half: everything that reads or advances the streams' state, and the render into out_all / out (defer: splines
        and footprints only -- self._deferred / the single stitchers' _deferred -- the push launches the render itself)."""
        st, S, e = self.static, self.S, 126
        ps, pt = st['pair_s'], st['pair_t']
        tm_out = (pt[0, 1], pt[1, 1])
        if off1 is None:
            off1, off_ref, off_tgt = _spatial_temporal_heads(self.spatial, self.temporal, f64, st['prev_feat'], feat, S, tm_out,
                                                             chain=self.chain)
        else:
            off_ref, off_tgt = _heads_b(self.spatial, self.temporal, f64, off1, st['prev_feat'], feat, S, tm_out, self.chain)
        ops.spatial_meshes(off1, off_ref, off_tgt, pipeline.LR_H, pipeline.LR_W,
                           out=(ps[0, 1].view(S, 7, 9, 2), ps[1, 1].view(S, 7, 9, 2)))
        st['prev_feat'].copy_(feat)
        # tsmotion of all streams and both views as ONE batch of 4 S frames laid out [view][prev, new][stream]: frame k pairs
        # with frame k - S (its own stream's previous frame); the rows of the `prev` halves are computed and ignored
        ops.tsmotion(ps.view(4 * S, 7, 9, 2), pt.view(4 * S, 7, 9, 2), pipeline.LR_H, pipeline.LR_W,
                     out=(st['ts_out'][0], st['ts_out'][1]), lag=S)
        # shift the 4 S rings, append this push's rows (smesh v0, smesh v1, tsm v0, tsm v1 of every stream), and make the
        # current spatial motions the previous ones
        ops.window_push(st['ring'] if S > 1 else st['ring'].view(4, WINDOW, e), st['ts_out'],
                        [1 * S * e, 3 * S * e, (4 * S + 1 * S) * e, (4 * S + 3 * S) * e], state=ps,
                        blocks=2, block=S * e, stride=2 * S * e, delta=S * e, per=S)
        r = st['ring'].view(4, S * WINDOW, 7, 9, 2)
        outs, _ = self.smooth.run_windows(r[0], r[1], r[2], r[3], S, WINDOW, WINDOW, 1)       # S windows, one per stream
        m1, m2 = outs['smooth_mesh1'], outs['smooth_mesh2']                                    # [S,7,7,9,2]
        if self.meshes_only:
            self.last_meshes = (m1[:, -1:], m2[:, -1:])                                       # [S,1,7,9,2]
            return
        # every stream's newest smoothed mesh on its own canvas: one normalisation launch per view and ONE batched TPS solve for
        # the 2 S splines (a solve is latency-bound, ~48 us whether it holds 2 systems or 16), then the render stream by stream
        guard = self.single[0]._guard()
        watch = None
        if FUSED_SPLINES:                    # control points + splines in one launch; the watcher rides on the footprint launch(es)
            src, T = ops.stream_splines([m1[0, -1], m2[0, -1]], WINDOW * e, st['bboxes'], self.single[0].nrigid, self.h, self.w)
            if self.fusion_mode == 'AVERAGE' and pipeline.SKIP_OUTSIDE:
                watch = (guard, st['watch_i'], st['watch_f'])
            else:
                ops.canvas_watch(src, st['watch_i'], st['watch_f'], guard)
        else:
            src = ops.stream_normalize_watch([m1[0, -1], m2[0, -1]], WINDOW * e, st['bboxes'], self.h, self.w, guard,
                                             st['watch_i'], st['watch_f'])                                         # [S,2,63,2]
            T = ops.tps_solve_shared(src.view(2 * S, 63, 2), self.single[0].nrigid).view(S, 2, 2, 66)
        if out_all is not None:
            # all streams render onto canvases of ONE size (e.g. the caller fixed them): the S current frames are a clip
            hc, wc = self.single[0].hc, self.single[0].wc
            if self.fusion_mode == 'AVERAGE':
                fp = ops.render_footprints(src, T, self.h, self.w, hc, wc, watch=watch) if pipeline.SKIP_OUTSIDE else None
                if defer:
                    self._deferred = (src, T, fp)
                    return
                ops.render_average_clip([hr1, hr2], src, T, hc, wc, self.warp_mode, out=out_all, footprint=fp)
            else:
                ops.render_linear_clip([hr1, hr2], src, T, hc, wc, self.warp_mode, out=out_all)
            return
        for s, one in enumerate(self.single):
            one._render_solved(None if defer else hr1[s:s + 1], None if defer else hr2[s:s + 1], src[s], T[s], out=_DEFER if defer else out[s],
                               watch=None if watch is None else (guard, st['watch_i'][s:s + 1], st['watch_f'][s:s + 1]))

    def _push_static(self, hr1, hr2, lr1, lr2, u8=None):
        st = self.static
        if self.grow == 'recapture' and not self.meshes_only:
            self._poll_growth()
        if self.trunk_pair is not None and self.trunk_versions != self._versions():
            self.trunk_pair = None           # a net was reloaded / moved: restack the twin trunk and recapture
            self.graph = None
        direct = self._direct()
        if u8 is not None:               # push_u8: decoded [S,H,W,3] frames of both views -- the cv2-exact resize writes the LR inputs
            ops.ingest_u8(u8[0], pipeline.LR_H, pipeline.LR_W, want_hr=False, lr_out=st['lr1'])
            ops.ingest_u8(u8[1], pipeline.LR_H, pipeline.LR_W, want_hr=False, lr_out=st['lr2'])
        else:
            if self.chain:               # (pair s = views s, s + 1: lr1 holds views 0 .. S-1, lr2's last row is view S)
                st['lrc'][:self.S].copy_(lr1); st['lrc'][self.S:].copy_(lr2[self.S - 1:])
            else:
                st['lr1'].copy_(lr1); st['lr2'].copy_(lr2)
            if not self.meshes_only and not direct:
                st['hr1'].copy_(hr1); st['hr2'].copy_(hr2)
        if not self.use_graph:
            self._step_static()
        elif self.graph is None:
            keep = {k: st[k].clone() for k in self._STATE + (() if self.meshes_only else ('watch_i', 'watch_f'))}
            side = _warmup_stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._step_static()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            for k, v in keep.items():
                st[k].copy_(v)
            g = _new_graph()
            with torch.cuda.graph(g):
                self._step_static()
            self.graph = g
            self.graph_nodes = _graph_nodes(g)
            for k, v in keep.items():
                st[k].copy_(v)
            self.graph.replay()
        else:
            self.graph.replay()
        self.frames_in += 1
        if self.meshes_only:             # (m1, m2) [S,1,7,9,2]: copies of the step's own tensors
            return tuple(m.clone() for m in self.last_meshes)
        if self.grow == 'recapture':
            self._post_watch_copy()
        if direct and u8 is not None:    # uint8 frames in, uint8 video frames out: one clip-style launch, or one launch per canvas size
            if st['out_all'] is not None:
                src, T, fp = self._deferred
                one = self.single[0]
                frames = ops.render_average_clip_u8([u8[0], u8[1]], src, T, one.hc, one.wc, self.warp_mode, footprint=fp)
                return [[frames[s]] for s in range(self.S)]
            res = []
            for s, one in enumerate(self.single):
                src, T, fp = one._deferred
                res.append([ops.render_average_u8([u8[0][s], u8[1][s]], src, T, one.hc, one.wc, self.warp_mode, footprint=fp)])
            return res
        if direct:                       # the graph left splines and footprints: render from the caller's frames into new tensors
            hr1, hr2 = hr1.reshape(st['hr1'].shape), hr2.reshape(st['hr2'].shape)
            if st['out_all'] is not None:
                src, T, fp = self._deferred
                one = self.single[0]
                frames = ops.render_average_clip([hr1.contiguous(), hr2.contiguous()], src, T, one.hc, one.wc, self.warp_mode, footprint=fp)
                return [[frames[s]] for s in range(self.S)]
            res = []
            for s, one in enumerate(self.single):
                src, T, fp = one._deferred
                res.append([ops.render_average([hr1[s:s + 1], hr2[s:s + 1]], src, T, one.hc, one.wc, self.warp_mode, footprint=fp)])
            return res
        return [[o.clone()] for o in st['out']]

    @torch.no_grad()
    def push_u8(self, frames1, frames2):
        """One DECODED frame pair of every stream: uint8 [S,H,W,3] device tensors (cv2 layout) -> S lists of stitched video frames
        uint8 [Hc,Wc,3]; byte for byte ops.ingest_u8 -> push -> ops.canvas_to_u8 (see OnlineStitcher.push_u8)."""
        S = self.S
        if self.meshes_only or self.chain:
            raise ValueError('push_u8 renders frames of independent streams: not for meshes_only / chain stitchers')
        for f in (frames1, frames2):
            if f.dtype != torch.uint8 or f.dim() != 4 or f.shape[0] != S or f.shape[-1] != 3 or tuple(f.shape) != tuple(frames1.shape):
                raise ValueError('push_u8 takes two uint8 [%d,H,W,3] tensors' % S)
        with ops.deterministic(self.deterministic):
            if self.static is not None and self._direct() and type(self)._push_static is MultiOnlineStitcher._push_static:
                return self._push_static(None, None, None, None, u8=(frames1.contiguous(), frames2.contiguous()))
            hr1, lr1 = ops.ingest_u8(frames1, pipeline.LR_H, pipeline.LR_W)
            hr2, lr2 = ops.ingest_u8(frames2, pipeline.LR_H, pipeline.LR_W)
            outs = self._push(hr1, hr2, lr1, lr2)
            return [[ops.canvas_to_u8(f.reshape((1,) + tuple(f.shape[-3:])))[0] for f in per] for per in outs]

    @torch.no_grad()
    def push(self, hr1, hr2, lr1, lr2):
        """One frame pair of every stream: hr* [S,3,H,W] (0..255), lr* [S,3,360,480] ([-1,1]), device tensors.
        -> S lists of newly stitched frames (empty for the first 6 pushes, 7 frames each on the 7th, then one per push)."""
        S = self.S
        if lr1.shape[0] != S or lr2.shape[0] != S or (not self.meshes_only and (hr1.shape[0] != S or hr2.shape[0] != S)):
            raise ValueError('expected %d streams per push' % S)          # (meshes_only: hr1 / hr2 are not looked at, None will do)
        with ops.deterministic(self.deterministic):
            return self._push(hr1, hr2, lr1, lr2)

    def _push(self, hr1, hr2, lr1, lr2):
        S = self.S
        if self.static is not None:
            return self._push_static(hr1, hr2, lr1, lr2)
        hs = (lambda h, s: None) if self.meshes_only else (lambda h, s: h[s:s + 1])
        outs = [one.push(hs(hr1, s), hs(hr2, s), lr1[s:s + 1], lr2[s:s + 1]) for s, one in enumerate(self.single)]
        self.frames_in += 1
        if self.single[0].static is not None:      # every stream's first window is complete: switch to the batched step
            self._init_static()
        if self.meshes_only:
            if outs[0] is None:
                return None
            self.last_meshes = (torch.stack([o[0] for o in outs], 0), torch.stack([o[1] for o in outs], 0))   # [S,7,7,9,2]
            return self.last_meshes
        return outs


class PipelinedMultiOnlineStitcher(_TwoInFlight, MultiOnlineStitcher):
    """MultiOnlineStitcher (S live pairs advancing together) with TWO pushes in flight (round 6; opt-in; see _TwoInFlight): every
    stream's frames bit-identical to MultiOnlineStitcher's, handed out one push late:
        st = PipelinedMultiOnlineStitcher(nets, H, W, streams=8)
        for batch in source: per_stream = st.push(*batch)        # S lists: [] x 6, 7 frames, [] (one push of lag), then 1 frame each
        per_stream = st.flush()
    Canvases are fixed after the first window (grow='never')."""

    def __init__(self, nets, height, width, streams, canvases=None, margin=0.03, warp_mode='NORMAL', fusion_mode='AVERAGE',
                 deterministic=False):
        MultiOnlineStitcher.__init__(self, nets, height, width, streams, canvases, margin, warp_mode, fusion_mode, use_graph=True,
                                     grow='never', meshes_only=False, deterministic=deterministic)
        if not L.QUAD:
            raise ValueError('PipelinedMultiOnlineStitcher needs the shared regressor launches (SS_QUAD_REGRESSOR=1)')
        self._pipe_init()

    def _pipe_alloc(self):
        d, S = self.dev, self.S
        two = lambda *shape: [torch.empty(shape, device=d) for _ in range(2)]
        fh, fw = pipeline.LR_H // 8, pipeline.LR_W // 8
        P = {'lr': two(2, S, 3, pipeline.LR_H, pipeline.LR_W),
             'hr': [[None, None], [None, None]] if self._direct() else two(2, S, 3, self.h, self.w),     # (direct render: no copies)
             'f64': two(2 * S, fh, fw, 128), 'feat': two(2 * S, fh, fw, 128), 'off1': two(S, 8)}
        if self.static['out_all'] is not None:
            P['out_all'] = two(*self.static['out_all'].shape)
            P['out'] = [[oa[s] for s in range(S)] for oa in P['out_all']]
        else:
            P['out_all'] = [None, None]
            P['out'] = [[torch.empty_like(o) for o in self.static['out']] for _ in range(2)]
        return P

    def _pipe_state(self):
        return [self.static[k] for k in self._STATE + ('watch_i', 'watch_f')]

    def _pipe_empty(self):
        return [[] for _ in range(self.S)]

    def _pipe_load(self, p, sa, sb, hr1, hr2, lr1, lr2):
        P = self.pipe
        with torch.cuda.stream(sa):
            P['lr'][p][0].copy_(lr1); P['lr'][p][1].copy_(lr2)
        if not self._direct():               # (direct render: the second stream's render reads the caller's frames)
            with torch.cuda.stream(sb):
                P['hr'][p][0].copy_(hr1); P['hr'][p][1].copy_(hr2)
        for t in (hr1, hr2):
            t.record_stream(sb)
        for t in (lr1, lr2):
            t.record_stream(sa)

    def _run_a(self, p):
        P = self.pipe
        f64, feat, off1 = self._stage_a([P['lr'][p][0], P['lr'][p][1]])
        P['f64'][p].copy_(f64)
        P['feat'][p].copy_(feat)
        P['off1'][p].copy_(off1)

    def _run_b(self, p):
        P = self.pipe
        self._stage_b(P['f64'][p], P['feat'][p], P['off1'][p], P['hr'][p][0], P['hr'][p][1], P['out_all'][p], P['out'][p],
                      defer=self._direct())

    def _pipe_deferred(self):
        if self.static['out_all'] is not None:
            return getattr(self, '_deferred', None)
        return [getattr(one, '_deferred', None) for one in self.single]

    def _pipe_take(self, p, hr1, hr2, lr1, lr2):
        P = self.pipe
        if not self._direct():
            return [[o.clone()] for o in P['out'][p]]
        hr1, hr2 = hr1.reshape(self.S, 3, self.h, self.w), hr2.reshape(self.S, 3, self.h, self.w)
        if self.static['out_all'] is not None:
            src, T, fp = P['deferred'][p]
            one = self.single[0]
            frames = ops.render_average_clip([hr1.contiguous(), hr2.contiguous()], src, T, one.hc, one.wc, self.warp_mode, footprint=fp)
            return [[frames[s]] for s in range(self.S)]
        res = []
        for s, one in enumerate(self.single):
            src, T, fp = P['deferred'][p][s]
            res.append([ops.render_average([hr1[s:s + 1], hr2[s:s + 1]], src, T, one.hc, one.wc, self.warp_mode, footprint=fp)])
        return res

    def _push_static(self, hr1, hr2, lr1, lr2):
        return self._push_pipelined(hr1, hr2, lr1, lr2)


class ThreeViewOnlineStitcher:
    """Streaming form of the three-view script (test_online_tra_threeview.py:154-505, whose frame loops run the same sliding
    windows as the two-view script): one frame TRIPLE per push.  Two pair chains -- (view 1, view 2) and (view 2, view 3), run as ONE
    batch of two streams (`MultiOnlineStitcher(streams=2, meshes_only=True, chain=True)`: ring buffers, cached TemporalNet features,
    sliding SmoothNet windows; every launch serves both pairs, and the middle view passes the trunks ONCE per push: three image
    passes, not four) -- deliver the newest smoothed meshes; the composition (mesh alignment, middle plane, TPS re-projection of the
    outer views: threeview:345-420) and the three-image render (:421-505) run per frame on two FIXED boxes: the composition's "first
    canvas" (the box the reference takes over all frames of the aligned meshes: a normalisation frame, nothing is cropped by it) and
    the output canvas.  Both are fixed when the first window is complete (its 7 frames' boxes grown by `margin`) or given by the
    caller; with the offline boxes passed in the stream reproduces the offline frames (tests/test_gpu_round5.py, against the CPU
    oracle at 720p: tests/test_gpu_round6.py).  The steady state -- both chains, composition, render -- is ONE HIP graph.

        st = ThreeViewOnlineStitcher(nets, H, W)
        for frame in st.push(hr1, hr2, hr3, lr1, lr2, lr3): ...      # [], ..., 7 frames on the 7th push, then 1: [3,Hc,Wc] fp32

    Overflow of the fixed output canvas is watched as in OnlineStitcher (`clipped_frames`, `overflow_report()`); grow='recapture'
    re-fixes the OUTPUT canvas (and captures the graph again) when a mesh comes within half the margin of its edge."""

    def __init__(self, nets, height, width, canvas=None, first_canvas=None, margin=0.03, warp_mode='NORMAL', fusion_mode='AVERAGE',
                 use_graph=True, grow='never'):
        if grow not in ('never', 'recapture'):
            raise ValueError("grow must be 'never' or 'recapture'")
        self.grow = grow
        self.nets = nets
        self.dev = next(nets[0].parameters()).device
        self.h, self.w = height, width
        self.margin = margin
        self.warp_mode, self.fusion_mode = warp_mode, fusion_mode
        self.use_graph = use_graph
        # the two pair chains as a batch of two streams over the CHAIN of three views: every launch serves both pairs, view 2's
        # trunk features are computed once
        self.chains = MultiOnlineStitcher(nets, height, width, streams=2, margin=margin, warp_mode=warp_mode, fusion_mode=fusion_mode,
                                          use_graph=False, meshes_only=True, chain=True)
        box = lambda b: None if b is None else torch.tensor(b, dtype=torch.float32, device=self.dev)
        self.bbox, self.first_canvas = box(canvas), box(first_canvas)
        self.hc = self.wc = None
        self.nrigid = get_norm_mesh(get_rigid_mesh(1, height, width, device=self.dev), height, width).contiguous()
        self.ring_hr = []
        self.frames_in = 0
        self.static = None
        self.graph = None
        self.graph_nodes = None
        self.versions = None
        # overflow state of the output canvas (the methods are OnlineStitcher's)
        self.canvas_epoch = 0
        self.watch_i = self.watch_f = None
        self._watch_totals = [0, 0, -1]
        self._host_watch = self._host_event = None
        self._near_handled = 0

    # ------------------------------------------------------------------ overflow / growth of the output canvas: as OnlineStitcher
    _guard = OnlineStitcher._guard
    _set_canvas = OnlineStitcher._set_canvas

    def _direct(self):
        return bool(DIRECT_RENDER and self.use_graph and self.fusion_mode == 'AVERAGE')

    _needed_bbox = OnlineStitcher._needed_bbox
    _regrow = OnlineStitcher._regrow
    _poll_growth = OnlineStitcher._poll_growth
    _post_watch_copy = OnlineStitcher._post_watch_copy
    overflow_report = OnlineStitcher.overflow_report
    clipped_frames = OnlineStitcher.clipped_frames

    # ------------------------------------------------------------------ composition + render of k frames
    def _compose(self, m12, m23):
        """Pair meshes (m1, m2) [k,7,9,2] of both chains -> (mesh1, middle, mesh3) [1,k,7,9,2] in first-canvas pixels."""
        k = m12[0].shape[0]
        sh = lambda m: m.reshape(1, k, 7, 9, 2)
        return pipeline.three_view_compose(sh(m12[0]), sh(m12[1]), sh(m23[0]), sh(m23[1]), self.h, self.w,
                                           first_canvas=self.first_canvas)

    def _compose_render(self, m1, m2, imgs, out):
        """One steady-state triple from the chains' newest meshes m1, m2 [2,1,7,9,2] (stream 0 = pair (1,2), stream 1 = pair (2,3)):
        composition + the render's splines as ONE launch (FUSED_SPLINES; else the seven launches it replaces), then the render."""
        if FUSED_SPLINES:
            meshes, src, T = ops.three_view_splines(m1[0], m2[0], m1[1], m2[1], self.first_canvas, self.bbox, self.nrigid, self.h, self.w)
            self.last_composed = meshes
            return self._render(imgs, meshes, out=out, splines=(src[0], T[0]))
        meshes = self._compose((m1[0], m2[0]), (m1[1], m2[1]))
        self.last_composed = meshes
        return self._render(imgs, meshes, out=out)

    def _grown(self, bb):
        bb = bb.cpu()
        gw, gh = self.margin * (bb[1] - bb[0]), self.margin * (bb[3] - bb[2])
        return torch.stack((bb[0] - gw, bb[1] + gw, bb[2] - gh, bb[3] + gh)).to(self.dev)

    def _render(self, imgs, meshes, out=None, splines=None):
        """imgs: three [1,3,H,W]; meshes: (mesh1, middle, mesh3) [1,1,7,9,2] in first-canvas pixels -> [3,Hc,Wc].
        splines = (src [3,63,2], T [3,2,66]) from ops.three_view_splines: the steady-state push has them already, and the overflow
        watcher runs inside the footprint launch (or as a launch of its own when there is none)."""
        watch = None
        if splines is None:
            src4 = ops.stream_normalize_watch([m.contiguous() for m in meshes], 126, self.bbox, 0.0, 0.0, self._guard(), self.watch_i,
                                              self.watch_f)                          # [1,3,63,2] on the output canvas
            src = src4[0]
            T = ops.tps_solve_shared(src, self.nrigid)
        else:
            src, T = splines
            if self.fusion_mode == 'AVERAGE' and pipeline.SKIP_OUTSIDE:
                watch = (self._guard(), self.watch_i, self.watch_f)
            else:
                ops.canvas_watch(src[None], self.watch_i, self.watch_f, self._guard())
        if self.fusion_mode == 'AVERAGE':
            fp = ops.render_footprints(src[None], T[None], self.h, self.w, self.hc, self.wc, watch=watch)[0] if pipeline.SKIP_OUTSIDE else None
            if out is _DEFER:
                self._deferred = (src, T, fp)
                return None
            return ops.render_average(imgs, src, T, self.hc, self.wc, self.warp_mode, out=out, footprint=fp)
        w = ops.tps_warp_views(imgs, src, T, self.hc, self.wc, self.warp_mode)        # [3,4,Hc,Wc]
        f = ops.linear_blend(w[0, 0:3], w[1, 0:3], w[0, 3], w[1, 3])
        res = ops.linear_blend(f, w[2, 0:3], ops.mask_union(w[0, 3], w[1, 3]), w[2, 3])
        return res if out is None else out.copy_(res)

    # ------------------------------------------------------------------ steady state
    def _versions(self):
        return tuple(n.weights_version for n in self.nets)

    def _step_static(self):
        ch, st = self.chains, self.static
        ch._step_static()
        m1, m2 = ch.last_meshes                                  # [2,1,7,9,2]: stream 0 = pair (1,2), stream 1 = pair (2,3)
        hr = st['hr']
        self._compose_render(m1, m2, [hr[0:1], hr[1:2], hr[2:3]], _DEFER if self._direct() else st['out'])

    def _state(self):
        return [self.chains.static[k] for k in MultiOnlineStitcher._STATE] + [self.watch_i, self.watch_f]

    def _push_static(self, hr1, hr2, hr3, lr1, lr2, lr3, u8=None):
        c, st = self.chains.static, self.static
        if self.grow == 'recapture':
            self._poll_growth()
        if self.versions != self._versions():         # a net was reloaded / moved: stale twin trunk and graph
            self.chains.trunk_pair = None
            self.graph = None
            self.versions = self._versions()
        direct = self._direct()
        for k, (h, l) in enumerate(((hr1, lr1), (hr2, lr2), (hr3, lr3))):      # each view once: three frames, three LR frames
            if u8 is not None:                                                 # decoded uint8 frames: resized into the graph's LR inputs
                ops.ingest_u8(u8[k][None], pipeline.LR_H, pipeline.LR_W, want_hr=False, lr_out=c['lrc'][k:k + 1])
                continue
            if not direct:                                                     # (direct: the render reads the caller's frames)
                st['hr'][k:k + 1].copy_(h.reshape(st['hr'][k:k + 1].shape))
            c['lrc'][k:k + 1].copy_(l.reshape(c['lrc'][k:k + 1].shape))
        if not self.use_graph:
            self._step_static()
        elif self.graph is None:
            state = self._state()
            keep = [t.clone() for t in state]
            side = _warmup_stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._step_static()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            for t, v in zip(state, keep):
                t.copy_(v)
            g = _new_graph()
            with torch.cuda.graph(g):
                self._step_static()
            self.graph = g
            self.graph_nodes = _graph_nodes(g)
            for t, v in zip(state, keep):          # capture does not execute: state is still the pre-push state
                t.copy_(v)
            self.graph.replay()
        else:
            self.graph.replay()
        self.frames_in += 1
        if self.grow == 'recapture':
            self._post_watch_copy()
        if direct:                       # the graph left splines and footprints; the render reads the caller's frames, writes a new tensor
            src, T, fp = self._deferred
            if u8 is not None:
                return [ops.render_average_u8(list(u8), src, T, self.hc, self.wc, self.warp_mode, footprint=fp)]
            shp = st['hr'][0:1].shape
            return [ops.render_average([hr1.reshape(shp), hr2.reshape(shp), hr3.reshape(shp)], src, T, self.hc, self.wc, self.warp_mode,
                                       footprint=fp)]
        return [st['out'].clone()]

    @torch.no_grad()
    def push_u8(self, img1, img2, img3):
        """One DECODED frame triple: uint8 [H,W,3] device tensors (cv2 layout) -> list of stitched video frames uint8 [Hc,Wc,3]; byte for
        byte ops.ingest_u8 -> push -> ops.canvas_to_u8 (see OnlineStitcher.push_u8)."""
        imgs = (img1, img2, img3)
        if any(i.dtype != torch.uint8 or i.dim() != 3 or i.shape[-1] != 3 or tuple(i.shape) != tuple(img1.shape) for i in imgs):
            raise ValueError('push_u8 takes three uint8 [H,W,3] frames')
        if self.static is not None and self._direct() and type(self)._push_static is ThreeViewOnlineStitcher._push_static:
            return self._push_static(None, None, None, None, None, None, u8=tuple(i.contiguous() for i in imgs))
        hr, lr = ops.ingest_u8(torch.stack(imgs, 0), pipeline.LR_H, pipeline.LR_W)
        frames = self.push(hr[0:1], hr[1:2], hr[2:3], lr[0:1], lr[1:2], lr[2:3])
        return [ops.canvas_to_u8(f.reshape((1,) + tuple(f.shape[-3:])))[0] for f in frames]

    @torch.no_grad()
    def push(self, hr1, hr2, hr3, lr1, lr2, lr3):
        """One frame triple: hr* [1,3,H,W] (0..255), lr* [1,3,360,480] ([-1,1]), device tensors.
        -> list of newly stitched frames (empty for the first 6 pushes, 7 frames on the 7th, then one per push)."""
        if self.static is not None:
            return self._push_static(hr1, hr2, hr3, lr1, lr2, lr3)
        sh = lambda t, c: t.reshape((1,) + tuple(c))
        got = self.chains.push(None, None, torch.cat((sh(lr1, lr1.shape[-3:]), sh(lr2, lr2.shape[-3:])), 0),
                               torch.cat((sh(lr2, lr2.shape[-3:]), sh(lr3, lr3.shape[-3:])), 0))
        self.ring_hr.append((hr1, hr2, hr3))
        self.frames_in += 1
        if got is None:
            return []
        m12, m23 = (got[0][0], got[1][0]), (got[0][1], got[1][1])                  # per chain: (m1, m2) [7,7,9,2]
        # first window complete: fix the first canvas and the output canvas, render its 7 frames, switch to the static step
        if self.first_canvas is None:
            k = m12[0].shape[0]
            shp = lambda m: m.reshape(1, k, 7, 9, 2)
            a1, a2, b1, b2, _ = ops.three_view_align(shp(m12[0]), shp(m12[1]), shp(m23[0]), shp(m23[1]), self.h, self.w)
            self.first_canvas = self._grown(ops.mesh_bbox([a1, a2, b1, b2], 0.0, 0.0))
        meshes = self._compose(m12, m23)                                           # three x [1,7,7,9,2]
        if self.bbox is None:
            self.bbox = self._grown(pipeline.canvas_bbox(meshes, self.h, self.w, prescaled=True))
        self._set_canvas()
        frames = [self._render(list(hr), [m[:, i:i + 1] for m in meshes]) for i, hr in enumerate(self.ring_hr)]
        self.ring_hr = []
        self.static = {'out': torch.empty((3, self.hc, self.wc), device=self.dev),
                       'hr': torch.empty((3, 3, self.h, self.w), device=self.dev)}
        self.versions = self._versions()
        return frames


class PipelinedThreeViewOnlineStitcher(_TwoInFlight, ThreeViewOnlineStitcher):
    """ThreeViewOnlineStitcher with TWO pushes in flight (round 6; opt-in; see _TwoInFlight): the three views' trunks and the two
    pairs' stage-1 heads of triple t + 1 run beside triple t's regressor heads, smoothing, composition and three-image render.  Frames
    bit-identical to ThreeViewOnlineStitcher's, handed out one push late (`flush()` for the last); boxes fixed after the first window."""

    def __init__(self, nets, height, width, canvas=None, first_canvas=None, margin=0.03, warp_mode='NORMAL', fusion_mode='AVERAGE'):
        ThreeViewOnlineStitcher.__init__(self, nets, height, width, canvas, first_canvas, margin, warp_mode, fusion_mode,
                                         use_graph=True, grow='never')
        if not L.QUAD:
            raise ValueError('PipelinedThreeViewOnlineStitcher needs the shared regressor launches (SS_QUAD_REGRESSOR=1)')
        self._pipe_init()

    # the twin trunk lives in the pair chains
    trunk_pair = property(lambda self: self.chains.trunk_pair, lambda self, v: setattr(self.chains, 'trunk_pair', v))
    trunk_versions = property(lambda self: self.chains.trunk_versions, lambda self, v: setattr(self.chains, 'trunk_versions', v))

    def _pipe_alloc(self):
        d = self.dev
        two = lambda *shape: [torch.empty(shape, device=d) for _ in range(2)]
        fh, fw = pipeline.LR_H // 8, pipeline.LR_W // 8
        return {'lrc': two(3, 3, pipeline.LR_H, pipeline.LR_W), 'hr': [None, None] if self._direct() else two(3, 3, self.h, self.w),
                'f64': two(3, fh, fw, 128),
                'feat': two(3, fh, fw, 128), 'off1': two(2, 8), 'out': [None, None] if self._direct() else two(3, self.hc, self.wc)}

    def _pipe_state(self):
        return self._state()

    def _pipe_empty(self):
        return []

    def _pipe_load(self, p, sa, sb, hr1, hr2, hr3, lr1, lr2, lr3):
        P = self.pipe
        for k, (h, l) in enumerate(((hr1, lr1), (hr2, lr2), (hr3, lr3))):
            with torch.cuda.stream(sa):
                P['lrc'][p][k:k + 1].copy_(l.reshape(P['lrc'][p][k:k + 1].shape))
            if not self._direct():           # (direct render: the second stream's render reads the caller's frames)
                with torch.cuda.stream(sb):
                    P['hr'][p][k:k + 1].copy_(h.reshape(P['hr'][p][k:k + 1].shape))
            h.record_stream(sb)
            l.record_stream(sa)

    def _run_a(self, p):
        P = self.pipe
        f64, feat, off1 = self.chains._stage_a(P['lrc'][p])
        P['f64'][p].copy_(f64)
        P['feat'][p].copy_(feat)
        P['off1'][p].copy_(off1)

    def _run_b(self, p):
        P, ch = self.pipe, self.chains
        ch._stage_b(P['f64'][p], P['feat'][p], P['off1'][p], None, None, None, None)
        m1, m2 = ch.last_meshes                                  # [2,1,7,9,2]: stream 0 = pair (1,2), stream 1 = pair (2,3)
        hr = P['hr'][p]
        imgs = None if hr is None else [hr[0:1], hr[1:2], hr[2:3]]
        self._compose_render(m1, m2, imgs, _DEFER if self._direct() else P['out'][p])

    def _pipe_take(self, p, hr1, hr2, hr3, lr1, lr2, lr3):
        P = self.pipe
        if self._direct():
            src, T, fp = P['deferred'][p]
            shp = (1, 3, self.h, self.w)
            return [ops.render_average([hr1.reshape(shp), hr2.reshape(shp), hr3.reshape(shp)], src, T, self.hc, self.wc, self.warp_mode,
                                       footprint=fp)]
        return [P['out'][p].clone()]

    def _push_static(self, hr1, hr2, hr3, lr1, lr2, lr3):
        return self._push_pipelined(hr1, hr2, hr3, lr1, lr2, lr3)


class HostFrameStream:
    """The reference's frame loop end to end, one pair (or triple) at a time: decoded uint8 frames in HOST memory in, stitched uint8
    video frames in pinned host memory out (test_online_tra.py:250-278 reads and resizes a frame pair per iteration, :409-417 writes
    the fused frame) -- the PCIe copies of neighbouring pushes hidden behind the current push on three HIP streams:

        st = OnlineStitcher(nets, H, W)                            # or ThreeViewOnlineStitcher
        for frame in HostFrameStream(st).run(source):              # source yields (img1, img2[, img3]) uint8 [H,W,3] host frames
            writer.write(frame.numpy())                            # uint8 [Hc,Wc,3], pinned host memory

    The upload of pair t + prefetch is enqueued before pair t's launches (`stitcher.push_u8` on the compute stream: the resize feeds the
    graph, the render samples the uploaded uint8 frames and writes the uint8 frame); the download of push t - 1's frame is enqueued
    behind push t's launches and runs beside them.  Which copy may start is decided on the HOST, not by GPU-side waits of a copy stream
    on a compute event: with pushes of 0.8 ms those waits cost 0.13 - 0.17 ms each and serialised copies and compute (1.08 ms per push;
    `tools/diag_host_stream2.py`) -- the host waits for push t - 1's end event while push t is already queued, and a staging slot is reused
    only when its last reader is known to have ended.  Frames come out in order, `depth` frames late at most; a yielded tensor stays
    valid until `depth` more frames have been yielded.  Pinned (page-locked) input frames upload asynchronously; pageable ones (numpy
    arrays) work and stall the loop for their copy."""

    def __init__(self, stitcher, depth=4, prefetch=2, streams=None):
        self.st, self.dev = stitcher, stitcher.dev
        self.depth, self.prefetch = int(depth), max(1, int(prefetch))
        self.up, self.comp, self.down = streams if streams is not None else pipeline.io_streams(self.dev)
        self._in, self._free, self._k = None, None, 0          # ring of prefetch + 3 uploaded pairs and the end events of their readers
        self._host, self._hk = None, 0                          # ring of 2 depth + 9 pinned result frames (7 arrive at once)

    def _stage(self, frames):
        src = [torch.from_numpy(f) if not torch.is_tensor(f) else f for f in frames]
        n = self.prefetch + 3
        if self._in is None or tuple(self._in[0][0].shape) != tuple(src[0].shape) or len(self._in[0]) != len(src):
            torch.cuda.synchronize(self.dev)
            self._in = [[torch.empty(tuple(t.shape), dtype=torch.uint8, device=self.dev) for t in src] for _ in range(n)]
            self._free = [None] * n
        j = self._k % n
        self._k += 1
        if self._free[j] is not None:
            self._free[j].synchronize()          # host-side: the push that read this slot ended long ago (returns at once)
        with torch.cuda.stream(self.up):
            for dst, t in zip(self._in[j], src):
                dst.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.up)
        return j, ev

    def _host_slot(self, shape):
        n = 2 * self.depth + 9
        if self._host is None or tuple(self._host[0].shape) != tuple(shape):
            torch.cuda.synchronize(self.dev)
            self._host = [torch.empty(tuple(shape), dtype=torch.uint8).pin_memory() for _ in range(n)]
        h = self._host[self._hk % n]
        self._hk += 1
        return h

    def _download(self, item, results):
        """item = (end event of a push, its frames): wait for the push on the HOST, then copy its frames down on the download stream."""
        done, outs = item
        if not outs:
            return
        done.synchronize()
        with torch.cuda.stream(self.down):
            for o in outs:
                h = self._host_slot(o.shape)
                h.copy_(o, non_blocking=True)
                o.record_stream(self.down)
                e = torch.cuda.Event()
                e.record(self.down)
                results.append((e, h))

    @torch.no_grad()
    def run(self, source):
        from collections import deque
        it = iter(source)
        staged, results = deque(), deque()
        prev = None                                          # the previous push: (end event, frames still on the device)
        for _ in range(self.prefetch):
            nxt = next(it, None)
            if nxt is not None:
                staged.append(self._stage(nxt))
        while staged:
            j, ev = staged.popleft()
            nxt = next(it, None)
            if nxt is not None:
                staged.append(self._stage(nxt))              # a later pair's upload is enqueued before this pair's launches
            with torch.cuda.stream(self.comp):
                self.comp.wait_event(ev)
                outs = self.st.push_u8(*self._in[j])
                done = torch.cuda.Event()
                done.record(self.comp)
            self._free[j] = done
            if prev is not None:
                self._download(prev, results)                # push t - 1's frames go down while push t (queued above) computes
            prev = (done, outs)
            while len(results) > self.depth:
                e, h = results.popleft()
                e.synchronize()
                yield h
        if prev is not None:
            self._download(prev, results)
        if hasattr(self.st, 'flush_u8'):                     # a pipelined stitcher still holds its last push's frame
            with torch.cuda.stream(self.comp):
                outs = self.st.flush_u8()
                done = torch.cuda.Event()
                done.record(self.comp)
            self._download((done, outs), results)
        while results:
            e, h = results.popleft()
            e.synchronize()
            yield h
