"""SmoothNet on the MI355X HIP engine; module API of Full_model_inference/Codes/smooth_network.py
(`SmoothNet()`, `build_SmoothNet(net, ts1, ts2, sm1, sm2)`, 14 state-dict tensors)."""
import torch
import torch.nn as nn

from . import grid_res, layers as L, ops

grid_h = grid_res.GRID_H
grid_w = grid_res.GRID_W
WINDOW_CHUNK = 2048          # sliding windows per SmoothNet pass ([2048,7,7,9,128] fp32 = 0.46 GB per activation)


class MotionPrediction(nn.Module):
    """Parameter container (smooth_network.py:106-137); embedding2 is in the checkpoint but never run."""

    def __init__(self, kernel=5):
        super().__init__()
        self.embedding1 = nn.Sequential(nn.Linear(2, 32), nn.ReLU())
        self.embedding2 = nn.Sequential(nn.Linear(1, 8), nn.ReLU())
        self.embedding3 = nn.Sequential(nn.Linear(2, 32), nn.ReLU())
        self.pad = kernel // 2
        self.MotionConv3D = nn.Sequential(
            nn.Conv3d(128, 128, (kernel, 3, 3), padding=(self.pad, 1, 1)), nn.ReLU(),
            nn.Conv3d(128, 128, (kernel, 3, 3), padding=(self.pad, 1, 1)), nn.ReLU(),
            nn.Conv3d(128, 128, (kernel, 3, 3), padding=(self.pad, 1, 1)), nn.ReLU())
        self.decoding = nn.Sequential(nn.Linear(128, 4))


class SmoothNet(L.PreparedMixin, nn.Module):
    def __init__(self, dropout=0.):
        super().__init__()
        self.MotionPre = MotionPrediction()
        self.eval()

    def _prepare(self):
        mp = self.MotionPre
        return {'e1': L.pack_fc(mp.embedding1[0]), 'e3': L.pack_fc(mp.embedding3[0]),
                'conv': [L.pack_conv3d(mp.MotionConv3D[i]) for i in (0, 2, 4)], 'pad': mp.pad,
                'dec': L.pack_fc(mp.decoding[0])}

    @torch.no_grad()
    def window_deltas(self, smesh1, smesh2, ts1, ts2, nw, t, wstride, zero_first, out=None):
        """The decoder output of `nw` windows: smesh*/ts* [frames,7,9,2] device tensors (see ss_smooth_embed for the window
        addressing) -> delta [nw,t,7,9,4].  Long clips run in chunks of WINDOW_CHUNK windows (one Conv3d launch addresses
        < 2 GiB of input) that write into one preallocated tensor."""
        p = self._prepared()
        if out is None:
            out = torch.empty((nw, t, grid_h + 1, grid_w + 1, 4), device=smesh1.device, dtype=torch.float32)
        if nw > WINDOW_CHUNK and wstride <= t:
            # window wi starts at frame wi * wstride, so a chunk is the same call on the frame range it covers
            for s in range(0, nw, WINDOW_CHUNK):
                m = min(WINDOW_CHUNK, nw - s)
                f0, f1 = s * wstride, (s + m - 1) * wstride + t
                self.window_deltas(smesh1[f0:f1], smesh2[f0:f1], ts1[f0:f1], ts2[f0:f1], m, t, wstride, zero_first, out[s:s + m])
            return out
        hid = ops.smooth_embed(smesh1, smesh2, ts1, ts2, p['e1'][0], p['e1'][1], p['e3'][0], p['e3'][1], nw, t,
                               wstride, zero_first)
        for w, b in p['conv']:
            hid = ops.conv(hid, w, b, stride=1, pad=(p['pad'], 1, 1), relu=True)
        ops.linear(hid.view(-1, 128), p['dec'][0], p['dec'][1], out=out.view(-1, 4))
        return out

    @torch.no_grad()
    def run_windows(self, smesh1, smesh2, ts1, ts2, nw, t, wstride, zero_first):
        """-> (dict of the 8 build_SmoothNet tensors, each [nw,t,7,9,2]; decoder output [nw,t,7,9,4])."""
        delta = self.window_deltas(smesh1, smesh2, ts1, ts2, nw, t, wstride, zero_first)
        out = ops.smooth_finalize(smesh1, smesh2, ts1, ts2, delta.view(-1, 4), nw, t, wstride, zero_first)
        return out, delta

    def forward(self, smesh_list1, smesh_list2, tsmotion_list1, tsmotion_list2):
        """smooth_network.py:64-101 -> (smesh1, smesh2, tsflow1, tsflow2, delta1, delta2), each [B,T,7,9,2]."""
        dev = next(self.parameters()).device
        t = len(smesh_list1)
        st = [torch.stack([x.to(dev).float() for x in lst], 1).contiguous()
              for lst in (smesh_list1, smesh_list2, tsmotion_list1, tsmotion_list2)]
        b = st[0].shape[0]
        flat = [x.view(b * t, grid_h + 1, grid_w + 1, 2) for x in st]
        o, delta = self.run_windows(flat[0], flat[1], flat[2], flat[3], b, t, t, 0)
        return (o['ori_mesh1'], o['ori_mesh2'], o['ori_path1'], o['ori_path2'], delta[..., 0:2], delta[..., 2:4])


@torch.no_grad()
def build_SmoothNet(net, tsmotion_list1, tsmotion_list2, smesh_list1, smesh_list2):
    """smooth_network.py:23-40 (note the builder's argument order: ts, ts, smesh, smesh)."""
    dev = next(net.parameters()).device
    t = len(smesh_list1)
    st = [torch.stack([x.to(dev).float() for x in lst], 1).contiguous()
          for lst in (smesh_list1, smesh_list2, tsmotion_list1, tsmotion_list2)]
    b = st[0].shape[0]
    flat = [x.view(b * t, grid_h + 1, grid_w + 1, 2) for x in st]
    return net.run_windows(flat[0], flat[1], flat[2], flat[3], b, t, t, 0)[0]
