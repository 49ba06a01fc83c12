"""SpatialNet / TemporalNet / SmoothNet (oracle, CPU, stock ATen ops).

State-dict key layout is the reference checkpoint layout, so the
same `state_dict` loads into the reference modules, these oracle modules and the
HIP-backed modules of `stabstitch2_amd`.

Reference sites (relative to /root/reference/Full_model_inference/Codes):
  ResNet-18 trunk slices     spatial_network.py:123-139 (torchvision 0.14.1 resnet18, restated)
  SpatialNet                 spatial_network.py:142-331
  cost volume                spatial_network.py:333-358, temporal_network.py:149-174
  CCL                        spatial_network.py:361-425
  build_SpatialNet           spatial_network.py:63-118
  TemporalNet / builder      temporal_network.py:23-34, 60-147
  SmoothNet / builder        smooth_network.py:23-157
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import GRID_H, GRID_W
from . import geometry as G
from . import samplers as S

NV = (GRID_H + 1) * (GRID_W + 1)


# --------------------------------------------------------------------------- trunk
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                            nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


def _layer(cin, cout, stride):
    return nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1))


def make_trunk():
    """(stage1, stage2) with the index layout of the reference's nn.Sequential slices."""
    stage1 = nn.Sequential(
        nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
        nn.MaxPool2d(3, 2, 1), _layer(64, 64, 1), _layer(64, 128, 2))
    stage2 = nn.Sequential(_layer(128, 256, 2))
    return stage1, stage2


def _regress_convs(cin, widths):
    """pairs of (3x3 no-bias conv + ReLU) followed by 2x2 max-pool; index layout 0,2,(4),5,7,(9)..."""
    mods = []
    c = cin
    for wd in widths:
        mods += [nn.Conv2d(c, wd, 3, padding=1, bias=False), nn.ReLU(inplace=True),
                 nn.Conv2d(wd, wd, 3, padding=1, bias=False), nn.ReLU(inplace=True),
                 nn.MaxPool2d(2, 2)]
        c = wd
    return nn.Sequential(*mods)


def _regress_fc(fin, h1, h2, fout):
    return nn.Sequential(nn.Linear(fin, h1), nn.ReLU(inplace=True), nn.Linear(h1, h2),
                         nn.ReLU(inplace=True), nn.Linear(h2, fout))


# --------------------------------------------------------------------------- correlation ops
def cost_volume(x1, x2, search_range, norm=False):
    """cv[j*K+i, y, x] = leaky_relu_0.1( mean_c x1[c,y,x] * x2[c, y+j-r, x+i-r] ), zero outside
    (spatial_network.py:333-358; norm=True L2-normalises both maps over channels first, :335-337 -- the signature's
    default, which the inference path overrides with norm=False at both call sites)."""
    if norm:
        x1 = F.normalize(x1, p=2, dim=1)
        x2 = F.normalize(x2, p=2, dim=1)
    r = search_range
    k = 2 * r + 1
    b, c, h, w = x1.shape
    x2p = F.pad(x2, [r, r, r, r])
    planes = []
    for j in range(k):
        for i in range(k):
            planes.append((x1 * x2p[:, :, j:j + h, i:i + w]).mean(dim=1))
    return F.leaky_relu(torch.stack(planes, dim=1), 0.1)


def ccl(f1, f2, softmax_scale=10.0):
    """Contextual correlation layer -> [B,2,h,w] (ch0 = dx, ch1 = dy)."""
    b, c, h, w = f1.shape
    n1 = F.normalize(f1, p=2, dim=1)
    n2 = F.normalize(f2, p=2, dim=1)
    out = []
    ky = torch.arange(h * w, dtype=torch.float32).div(w, rounding_mode='floor').view(-1, 1, 1)
    kx = torch.arange(h * w, dtype=torch.float32).remainder(w).view(-1, 1, 1)
    py = torch.arange(h, dtype=torch.float32).view(1, h, 1)
    px = torch.arange(w, dtype=torch.float32).view(1, 1, w)
    for n in range(b):
        # every 3x3 (zero padded) patch of n2 becomes one matching filter [h*w, c, 3, 3]
        filt = F.unfold(n2[n:n + 1], 3, padding=1).reshape(c, 3, 3, h * w).permute(3, 0, 1, 2)
        match = F.conv2d(n1[n:n + 1], filt.contiguous(), padding=1)[0]  # [h*w, h, w]
        prob = F.softmax(match * softmax_scale, dim=0)
        fy = (prob * (ky - py)).sum(dim=0)
        fx = (prob * (kx - px)).sum(dim=0)
        out.append(torch.stack((fx, fy), dim=0))
    return torch.stack(out, dim=0)


# --------------------------------------------------------------------------- SpatialNet
class SpatialNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.regressNet1_part1 = _regress_convs(2, (64, 128, 128))
        self.regressNet1_part2 = _regress_fc(768, 512, 128, 8)
        self.regressNet2_part1_ref = _regress_convs(121, (64, 128, 128, 256))
        self.regressNet2_part2_ref = _regress_fc(1536, 1024, 512, NV * 2)
        self.regressNet2_part1_tgt = _regress_convs(121, (64, 128, 128, 256))
        self.regressNet2_part2_tgt = _regress_fc(1536, 1024, 512, NV * 2)
        self.feature_extractor_stage1, self.feature_extractor_stage2 = make_trunk()

    def forward(self, in1, in2):
        b, _, ih, iw = in1.shape
        f1_64 = self.feature_extractor_stage1(in1)
        f1_32 = self.feature_extractor_stage2(f1_64)
        f2_64 = self.feature_extractor_stage1(in2)
        f2_32 = self.feature_extractor_stage2(f2_64)

        corr = ccl(f1_32, f2_32)
        offset_1 = self.regressNet1_part2(self.regressNet1_part1(corr).reshape(b, -1))

        _, H_tgt, H_ref = G.decompose(offset_1, ih, iw, scale=8.0)
        fh, fw = int(ih / 8), int(iw / 8)
        M = torch.tensor([[iw / 8 / 2.0, 0.0, iw / 8 / 2.0],
                          [0.0, ih / 8 / 2.0, ih / 8 / 2.0],
                          [0.0, 0.0, 1.0]])
        Minv = torch.inverse(M)
        th_ref = torch.matmul(torch.matmul(Minv, H_ref), M)
        th_tgt = torch.matmul(torch.matmul(Minv, H_tgt), M)
        w1 = S.homography_warp(f1_64, th_ref, (fh, fw))
        w2 = S.homography_warp(f2_64, th_tgt, (fh, fw))

        cv_ref = cost_volume(w1, w2, 5)
        offset_2_ref = self.regressNet2_part2_ref(self.regressNet2_part1_ref(cv_ref).reshape(b, -1))
        cv_tgt = cost_volume(w2, w1, 5)
        offset_2_tgt = self.regressNet2_part2_tgt(self.regressNet2_part1_tgt(cv_tgt).reshape(b, -1))
        return offset_1, offset_2_ref, offset_2_tgt


def build_SpatialNet(net, in1, in2):
    b, _, ih, iw = in1.shape
    offset_1, off_ref, off_tgt = net(in1, in2)
    _, H_tgt, H_ref = G.decompose(offset_1, ih, iw, scale=1.0)
    rigid = G.rigid_mesh(b, ih, iw)
    mesh_ref = G.homography_to_mesh(H_ref, rigid) + off_ref.reshape(b, GRID_H + 1, GRID_W + 1, 2)
    mesh_tgt = G.homography_to_mesh(H_tgt, rigid) + off_tgt.reshape(b, GRID_H + 1, GRID_W + 1, 2)
    return dict(motion1=mesh_ref - rigid, motion2=mesh_tgt - rigid)


# --------------------------------------------------------------------------- TemporalNet
class TemporalNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.regressNet2_part1 = _regress_convs(49, (64, 128, 128, 256))
        self.regressNet2_part2 = _regress_fc(1536, 1024, 512, NV * 2)
        self.feature_extractor_stage1, self.feature_extractor_stage2 = make_trunk()

    def forward(self, frames):
        motions = []
        prev = self.feature_extractor_stage1(frames[0])
        for t in range(1, len(frames)):
            cur = self.feature_extractor_stage1(frames[t])
            cv = cost_volume(prev, cur, 3)
            off = self.regressNet2_part2(self.regressNet2_part1(cv).reshape(cv.shape[0], -1))
            motions.append(off.reshape(-1, GRID_H + 1, GRID_W + 1, 2))
            prev = cur
        return motions


def build_TemporalNet(net, frames):
    b = frames[0].shape[0]
    motions = net(frames)
    motions.insert(0, torch.zeros(b, GRID_H + 1, GRID_W + 1, 2))
    return dict(motion_list=motions)


# --------------------------------------------------------------------------- SmoothNet
class MotionPrediction(nn.Module):
    def __init__(self, kernel=5):
        super().__init__()
        self.embedding1 = nn.Sequential(nn.Linear(2, 32), nn.ReLU())
        self.embedding2 = nn.Sequential(nn.Linear(1, 8), nn.ReLU())   # in the checkpoint, never run
        self.embedding3 = nn.Sequential(nn.Linear(2, 32), nn.ReLU())
        p = kernel // 2
        self.MotionConv3D = nn.Sequential(
            nn.Conv3d(128, 128, (kernel, 3, 3), padding=(p, 1, 1)), nn.ReLU(),
            nn.Conv3d(128, 128, (kernel, 3, 3), padding=(p, 1, 1)), nn.ReLU(),
            nn.Conv3d(128, 128, (kernel, 3, 3), padding=(p, 1, 1)), nn.ReLU())
        self.decoding = nn.Sequential(nn.Linear(128, 4))

    def forward(self, smesh1, smesh2, tsflow1, tsflow2):
        hid = torch.cat((self.embedding1(smesh1), self.embedding3(tsflow1),
                         self.embedding1(smesh2), self.embedding3(tsflow2)), dim=4)
        hid = self.MotionConv3D(hid.permute(0, 4, 1, 2, 3))
        return self.decoding(hid.permute(0, 2, 3, 4, 1))


class SmoothNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.MotionPre = MotionPrediction()

    def forward(self, smesh_list1, smesh_list2, tsmotion_list1, tsmotion_list2):
        def cum(lst):
            acc = [lst[0]]
            for t in range(1, len(lst)):
                acc.append(acc[-1] + lst[t])
            return torch.stack(acc, dim=1)   # [B,T,h,w,2]
        smesh1 = torch.stack(smesh_list1, dim=1)
        smesh2 = torch.stack(smesh_list2, dim=1)
        tsflow1 = cum(tsmotion_list1)
        tsflow2 = cum(tsmotion_list2)
        delta = self.MotionPre(smesh1, smesh2, tsflow1, tsflow2)
        return smesh1, smesh2, tsflow1, tsflow2, delta[..., 0:2], delta[..., 2:4]


def build_SmoothNet(net, tsmotion_list1, tsmotion_list2, smesh_list1, smesh_list2):
    om1, om2, op1, op2, d1, d2 = net(smesh_list1, smesh_list2, tsmotion_list1, tsmotion_list2)
    return dict(ori_path1=op1, smooth_path1=op1 + d1, ori_mesh1=om1, smooth_mesh1=om1 - d1,
                ori_path2=op2, smooth_path2=op2 + d2, ori_mesh2=om2, smooth_mesh2=om2 - d2)
