"""TemporalNet on the MI355X HIP engine; module API of Full_model_inference/Codes/temporal_network.py
(`TemporalNet()`, `build_TemporalNet(net, img_tensor_list)`, 104 state-dict tensors incl. the unused
feature_extractor_stage2).  The reference walks the clip frame by frame (temporal_network.py:129-145);
here every frame's stage-1 features are computed in one batched pass and all consecutive-pair cost
volumes / regressions in another (eval mode is batch invariant)."""
import torch
import torch.nn as nn

from . import grid_res, layers as L, ops

grid_h = grid_res.GRID_H
grid_w = grid_res.GRID_W


class TemporalNet(L.PreparedMixin, nn.Module):
    def __init__(self, dropout=0.):
        super().__init__()
        self.regressNet2_part1 = L.regress_convs(49, (64, 128, 128, 256))
        self.regressNet2_part2 = L.regress_fc(1536, 1024, 512, (grid_w + 1) * (grid_h + 1) * 2)
        self.feature_extractor_stage1, self.feature_extractor_stage2 = L.make_trunk()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
        self.eval()

    def _prepare(self):
        return {'s1': L.prep_trunk_stage1(self.feature_extractor_stage1),
                'r2': L.prep_regressor(self.regressNet2_part1, self.regressNet2_part2, 256, 6)}

    @torch.no_grad()
    def motions(self, frames):
        """frames [N,B,3,360,480] (device) -> [N-1,B,7,9,2] mesh motions between consecutive frames."""
        p = self._prepared()
        n, b = frames.shape[0], frames.shape[1]
        if n < 2:              # a single frame has no consecutive pair (the reference's loop body never runs)
            return torch.zeros((0, b, grid_h + 1, grid_w + 1, 2), device=frames.device, dtype=torch.float32)
        f = L.run_stage1(frames.reshape(n * b, *frames.shape[2:]), p['s1'])
        f = f.view(n, b, *f.shape[1:])
        x1 = f[:-1].reshape((n - 1) * b, *f.shape[2:])
        x2 = f[1:].reshape((n - 1) * b, *f.shape[2:])
        off = L.run_regressor(ops.cost_volume(x1, x2, 3), p['r2'])
        return off.view(n - 1, b, grid_h + 1, grid_w + 1, 2)

    @torch.no_grad()
    def motions_views(self, views):
        """views: list of V device tensors [N,3,360,480] (one clip per view) -> list of V tensors [N-1,7,9,2].
        All views share one trunk pass and one regressor pass; no copy of the input frames is made."""
        p = self._prepared()
        n = views[0].shape[0]
        v = len(views)
        f = L.run_stage1(list(views), p['s1'])                      # [V*N,45,60,128], view-major
        return self.motions_from_view_features([f[i * n:(i + 1) * n] for i in range(v)])

    @torch.no_grad()
    def motions_from_view_features(self, feats, zero_first=False):
        """feats: list of V nhwc tensors [N,45,60,128] (stage-1 features of every frame of a view)
        -> list of V tensors [N-1,7,9,2]; one regressor pass for all views.
        zero_first=True: -> V tensors [N,7,9,2] whose frame 0 is the zero motion (temporal_network.py:31-33), written in
        place (the last FC layer stores every view's motions behind its zero frame; no torch.cat)."""
        p = self._prepared()
        v, n = len(feats), feats[0].shape[0]
        cv = torch.empty((v * (n - 1), feats[0].shape[1], feats[0].shape[2], 52), device=feats[0].device,
                         dtype=torch.float32)
        for i in range(v):
            ops.cost_volume(feats[i][:n - 1], feats[i][1:], 3, out=cv[i * (n - 1):(i + 1) * (n - 1)])
        if not zero_first:
            off = L.run_regressor(cv, p['r2']).view(v, n - 1, grid_h + 1, grid_w + 1, 2)
            return [off[i] for i in range(v)]
        tm = torch.empty((v, n, grid_h + 1, grid_w + 1, 2), device=cv.device, dtype=torch.float32)
        for i in range(v):
            ops.fill(tm[i, 0])
        L.run_regressor(cv, p['r2'], out_slices=[(i * (n - 1), (i + 1) * (n - 1), tm[i, 1:].view(n - 1, -1)) for i in range(v)])
        return [tm[i] for i in range(v)]

    @torch.no_grad()
    def features(self, frames):
        """Stage-1 features of a list of [n_i,3,360,480] device tensors -> nhwc [sum n_i,45,60,128] (for callers that
        keep the previous frame's features, as the reference's loop does at temporal_network.py:144)."""
        return L.run_stage1(list(frames), self._prepared()['s1'])

    @torch.no_grad()
    def motions_from_features(self, f_prev, f_cur, out_slices=None):
        """nhwc features of consecutive frames [n,45,60,128] x2 -> mesh motions [n,7,9,2]
        (out_slices: [(row0, row1, dst [rows,126])] -- the regressor writes those rows there instead, returns None)."""
        off = L.run_regressor(ops.cost_volume(f_prev, f_cur, 3), self._prepared()['r2'], out_slices=out_slices)
        return None if out_slices is not None else off.view(-1, grid_h + 1, grid_w + 1, 2)

    def forward(self, img_tensor_list):
        dev = next(self.parameters()).device
        frames = torch.stack([t.to(dev, non_blocking=True).float() for t in img_tensor_list], 0)
        m = self.motions(frames)
        return [m[i] for i in range(m.shape[0])]

    @staticmethod
    def cost_volume(x1, x2, search_range, norm=True, fast=True):
        d = (2 * search_range + 1) ** 2
        a, b = ops.nchw_to_nhwc(x1), ops.nchw_to_nhwc(x2)
        if norm:
            a, b = ops.l2norm(a), ops.l2norm(b)
        return ops.nhwc_to_nchw(ops.cost_volume(a, b, search_range), d)


def build_TemporalNet(net, img_tensor_list):
    """temporal_network.py:23-34 -> dict(motion_list = [zeros] + N-1 motions), each [B,7,9,2]
    (a one-frame list gives [zeros], as the reference does)."""
    motion_list = net(img_tensor_list)
    dev = next(net.parameters()).device
    b = img_tensor_list[0].shape[0]
    motion_list.insert(0, torch.zeros((b, grid_h + 1, grid_w + 1, 2), device=dev, dtype=torch.float32))
    return dict(motion_list=motion_list)
