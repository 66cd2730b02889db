"""PoseCNN (reference networks/pose_cnn.py:9-45): seven stride-2 conv+ReLU stages, a 1x1 head, spatial
mean, x0.01, split into axis-angle and translation.  State-dict keys: pose_conv.*, net.{0..6}.* —
`pose_conv` is registered before `net`, as in the reference."""
import torch.nn as nn

from sqd import nnops as X

_STAGES = ((16, 7), (32, 5), (64, 3), (128, 3), (256, 3), (256, 3), (256, 3))


class PoseCNN(nn.Module):
    def __init__(self, num_input_frames):
        super().__init__()
        self.num_input_frames = num_input_frames
        convs, cin = [], 3 * num_input_frames
        for cout, k in _STAGES:
            convs.append(nn.Conv2d(cin, cout, k, 2, (k - 1) // 2))
            cin = cout
        self.pose_conv = nn.Conv2d(cin, 6 * (num_input_frames - 1), 1)
        self.num_convs = len(convs)
        self.net = nn.ModuleList(convs)

    def forward(self, out):
        return self._trunk(X.conv2d(out, self.net[0], "relu"))

    def forward_pairs(self, pairs):
        """pairs = [(first frame, second frame), ...], each [B,3,H,W] -> the outputs of forward(x) for the batch x [B * len(pairs), 6,
        H, W] whose row b * len(pairs) + i is torch.cat(pairs[i], 1)[b] (reference trainer.py:319-326), without building x"""
        return self._trunk(X.stem_pairs(pairs, self.net[0], "relu"))

    def _trunk(self, out):
        for conv in list(self.net)[1:]:
            out = X.conv2d(out, conv, "relu")
        # (axisangle, translation) = out.view(-1, F, 1, 6)[..., :3], [..., 3:] of the reference, written as two dense tensors
        return X.pose_head(out, self.pose_conv, 0.01, split=True)
