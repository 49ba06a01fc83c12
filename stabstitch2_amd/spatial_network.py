"""SpatialNet on the MI355X HIP engine; same module API as the reference
(Full_model_inference/Codes/spatial_network.py): `SpatialNet()`, `build_SpatialNet(net, a, b)`,
`H2Mesh`, `get_rigid_mesh`, `get_norm_mesh`, identical state-dict keys (130 tensors)."""
import torch
import torch.nn as nn

from . import grid_res, layers as L, ops

grid_h = grid_res.GRID_H
grid_w = grid_res.GRID_W


def get_rigid_mesh(batch_size, height, width, device=None):
    """spatial_network.py:39-50 -- [B,7,9,2] regular vertex grid in pixels (host-side constant)."""
    xs = torch.linspace(0.0, float(width), grid_w + 1)
    ys = torch.linspace(0.0, float(height), grid_h + 1)
    m = torch.stack((xs.view(1, -1).expand(grid_h + 1, -1), ys.view(-1, 1).expand(-1, grid_w + 1)), 2)
    m = m.unsqueeze(0).expand(batch_size, -1, -1, -1)
    return m.to(device) if device is not None else (m.cuda() if torch.cuda.is_available() else m)


def get_norm_mesh(mesh, height, width):
    """spatial_network.py:53-59 -- pixels -> [-1,1], flattened to [B,63,2] (torch elementwise glue)."""
    b = mesh.size()[0]
    x = mesh[..., 0] * 2. / float(width) - 1.
    y = mesh[..., 1] * 2. / float(height) - 1.
    return torch.stack([x, y], 3).reshape([b, -1, 2])


def H2Mesh(H, rigid_mesh):
    """spatial_network.py:20-36 -- mesh = persp_divide(H^-1 [x y 1]^T) (`ss_h2mesh`: the 3 x 3 inverse in fp64 on the device)."""
    b = rigid_mesh.shape[0]
    return ops.h2mesh(H, rigid_mesh).reshape(b, grid_h + 1, grid_w + 1, 2)


class SpatialNet(L.PreparedMixin, nn.Module):
    def __init__(self):
        super().__init__()
        nv2 = (grid_w + 1) * (grid_h + 1) * 2
        self.regressNet1_part1 = L.regress_convs(2, (64, 128, 128))
        self.regressNet1_part2 = L.regress_fc(768, 512, 128, 8)
        self.regressNet2_part1_ref = L.regress_convs(121, (64, 128, 128, 256))
        self.regressNet2_part2_ref = L.regress_fc(1536, 1024, 512, nv2)
        self.regressNet2_part1_tgt = L.regress_convs(121, (64, 128, 128, 256))
        self.regressNet2_part2_tgt = L.regress_fc(1536, 1024, 512, nv2)
        self.feature_extractor_stage1, self.feature_extractor_stage2 = L.make_trunk()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
        self.eval()

    def _prepare(self):
        return {
            's1': L.prep_trunk_stage1(self.feature_extractor_stage1),
            's2': L.prep_trunk_stage2(self.feature_extractor_stage2),
            'r1': L.prep_regressor(self.regressNet1_part1, self.regressNet1_part2, 128, 6),
            'r2_ref': L.prep_regressor(self.regressNet2_part1_ref, self.regressNet2_part2_ref, 256, 6),
            'r2_tgt': L.prep_regressor(self.regressNet2_part1_tgt, self.regressNet2_part2_tgt, 256, 6),
        }

    def _prepared(self):
        p = super()._prepared()
        if 'r2_pair' not in p:           # twin stage-2 regressors share every launch (grouped convs)
            p['r2_pair'] = L.pair_regressors(p['r2_ref'], p['r2_tgt'])
        return p

    @torch.no_grad()
    def forward(self, input1_tensor, input2_tensor):
        """[B,3,360,480] x2 in [-1,1] -> (offset_1 [B,8], offset_2_ref [B,126], offset_2_tgt [B,126])."""
        p = self._prepared()
        b, _, img_h, img_w = input1_tensor.shape
        f64 = L.run_stage1([input1_tensor, input2_tensor], p['s1'])      # [2B,45,60,128] nhwc
        return self.forward_features(f64, b, img_h, img_w)

    @torch.no_grad()
    def forward_features(self, f64, b, img_h, img_w):
        """Everything after the stage-1 trunk: f64 nhwc [2B,45,60,128] (view 1 first) -> the three offsets."""
        f32 = L.run_stage2(f64, self._prepared()['s2'])    # [2B,23,30,256]
        return self.forward_pair(f64[:b], f64[b:], f32[:b], f32[b:], img_h, img_w)

    @torch.no_grad()
    def trunk_features(self, x_list):
        """Stage-1 and stage-2 features of NCHW inputs (they depend on the image alone, so a view that takes part in two
        pairs -- the middle view of a three-view rig -- needs them once) -> (f64 [n,45,60,128], f32 [n,23,30,256])."""
        p = self._prepared()
        f64 = L.run_stage1(list(x_list), p['s1'])
        return f64, L.run_stage2(f64, p['s2'])

    @torch.no_grad()
    def forward_pair(self, f64_1, f64_2, f32_1, f32_2, img_h, img_w):
        """Both views' trunk features (each [B,...]) -> the three offsets (spatial_network.py:291-331)."""
        offset_1, cv = self.forward_pair_cv(f64_1, f64_2, f32_1, f32_2, img_h, img_w)
        offset_2_ref, offset_2_tgt = L.run_regressor_pair(cv, self._prepared()['r2_pair'])
        return offset_1, offset_2_ref, offset_2_tgt

    @torch.no_grad()
    def forward_pair_cv(self, f64_1, f64_2, f32_1, f32_2, img_h, img_w):
        """forward_pair up to the stage-2 cost volumes: -> (offset_1 [B,8], cv [2,B,h/8,w/8,124]: both directions).  The caller
        runs regressNet2 ref / tgt on cv -- alone (`forward_pair`) or together with TemporalNet's regressor in shared
        launches (layers.run_regressor_quad)."""
        offset_1 = self.offset1_from_features(f32_1, f32_2)
        return offset_1, self.cv_from_offset1(f64_1, f64_2, offset_1, img_h, img_w)

    @torch.no_grad()
    def offset1_from_features(self, f32_1, f32_2):
        """Stage 1 (spatial_network.py:291-300): contextual correlation of the 1/16 features -> global homography offsets [B,8]."""
        _, flow = ops.ccl(f32_1, f32_2, 10.0, want_nchw=False, want_nhwc4=True)
        return L.run_regressor(flow, self._prepared()['r1'])

    @torch.no_grad()
    def cv_from_offset1(self, f64_1, f64_2, offset_1, img_h, img_w):
        """Stage 2 up to its cost volumes (spatial_network.py:302-331): bidirectional decomposition at 1/8 scale, both feature maps
        warped onto the middle plane, local cost volumes in both directions -> [2,B,h/8,w/8,124] (one launch)."""
        th_ref, th_tgt = ops.spatial_decompose(offset_1, img_h, img_w)
        fh, fw = int(img_h / 8), int(img_w / 8)
        w1, w2 = ops.homo_warp_pair(f64_1, f64_2, th_ref, th_tgt, fh, fw)
        return ops.cost_volume_bidir(w1, w2, 5)

    @staticmethod
    def cost_volume(x1, x2, search_range, norm=True, fast=True):
        """Reference signature (NCHW in/out, spatial_network.py:333-358)."""
        d = (2 * search_range + 1) ** 2
        a, b = ops.nchw_to_nhwc(x1), ops.nchw_to_nhwc(x2)
        if norm:                 # F.normalize over channels first (the signature's default; inference passes norm=False)
            a, b = ops.l2norm(a), ops.l2norm(b)
        return ops.nhwc_to_nchw(ops.cost_volume(a, b, search_range), d)

    def CCL(self, feature_1, feature_2):
        """Reference signature (NCHW in/out, spatial_network.py:369-425)."""
        flow, _ = ops.ccl(ops.nchw_to_nhwc(feature_1), ops.nchw_to_nhwc(feature_2), 10.0, True, False)
        return flow


@torch.no_grad()
def build_SpatialNet(net, input1_tensor, input2_tensor):
    """spatial_network.py:63-118 -> dict(motion1, motion2), each [B,7,9,2] (mesh - rigid, LR px)."""
    _, _, img_h, img_w = input1_tensor.shape
    offset_1, off_ref, off_tgt = net(input1_tensor, input2_tensor)
    m1, m2 = ops.spatial_meshes(offset_1, off_ref, off_tgt, img_h, img_w)
    return dict(motion1=m1, motion2=m2)
