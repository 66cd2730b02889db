"""SQLdepth head ("Self Query Layer" decoder) for the MI355X build.

Surface kept from the reference: `Depth_Decoder_QueryTr(in_channels, embedding_dim, patch_size,
num_heads, query_nums, dim_out, norm, min_val, max_val)` with `forward(x0) -> {("disp", 0): depth}`
(reference networks/depth_decoder_QTR.py:7-74) and its ResNet-18 twin `Lite_Depth_Decoder_QueryTr`
(reference networks/lite_depth_decoder_QTR.py:7-72, feed-forward width 512 instead of 1024).
State-dict keys are those of SURVEY.md App. C.

Data flow: P x P patch embedding + learned positional code -> 4 post-norm transformer encoder layers
over the (H/2P)(W/2P) tokens -> the first Q tokens act as queries against every pixel of the 3x3
convolved feature map (FullQueryLayer: energy maps + softmax-over-pixels summaries) -> an MLP turns
the summaries into per-image adaptive bin widths -> a 1x1 conv + channel softmax over the energy
maps gives per-pixel bin probabilities -> expected bin centre = depth in (min_val, max_val)."""
import torch
import torch.nn as nn

from sqd import nnops as X

from .layers import FullQueryLayer


class _QueryTrBase(nn.Module):
    FEED_FORWARD = 1024

    def __init__(self, in_channels, embedding_dim=128, patch_size=16, num_heads=4, query_nums=100, dim_out=256,
                 norm="linear", min_val=0.001, max_val=10):
        super().__init__()
        self.norm = norm
        self.embedding_convPxP = nn.Conv2d(in_channels, embedding_dim, kernel_size=patch_size, stride=patch_size, padding=0)
        self.positional_encodings = nn.Parameter(torch.rand(500, embedding_dim), requires_grad=True)
        layer = nn.TransformerEncoderLayer(embedding_dim, num_heads, dim_feedforward=self.FEED_FORWARD)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=4, enable_nested_tensor=False)
        self.conv3x3 = nn.Conv2d(in_channels, embedding_dim, kernel_size=3, stride=1, padding=1)
        self.full_query_layer = FullQueryLayer()
        self.bins_regressor = nn.Sequential(nn.Linear(embedding_dim * query_nums, 16 * query_nums), nn.LeakyReLU(),
                                            nn.Linear(16 * query_nums, 16 * 16), nn.LeakyReLU(),
                                            nn.Linear(16 * 16, dim_out))
        self.convert_to_prob = nn.Sequential(nn.Conv2d(query_nums, dim_out, kernel_size=1, stride=1, padding=0),
                                             nn.Softmax(dim=1))
        self.query_nums = query_nums
        self.min_val = min_val
        self.max_val = max_val

    def forward(self, x0):
        # x0 has two consumers: the 3x3 convolution hands it through (x0p), so that the patch embedding's gradient is added in the
        # 3x3 data gradient's epilogue instead of by a pass of its own over [B,C,h,w] (same forward values: the order of two independent
        # convolutions)
        feat, x0p = X.conv2d(x0, self.conv3x3, skip=True)
        # embedding.flatten(2) + positional_encodings[:T].T, permuted to [T,B,E] (reference :49-51), one launch
        tokens = X.tokens_with_positions(X.conv2d(x0p, self.embedding_convPxP), self.positional_encodings)
        tokens = X.transformer_encoder(tokens, self.transformer_encoder)               # [T,B,E]
        queries = X.first_queries(tokens, self.query_nums)                             # first Q tokens, [B,Q,E]
        energy_maps, summaries = self.full_query_layer(feat, queries)
        bs, Q, E = summaries.shape
        y = summaries.reshape(bs, Q * E)
        y = X.linear(y, self.bins_regressor[0], "leaky_relu")
        y = X.linear(y, self.bins_regressor[2], "leaky_relu")
        y = X.linear(y, self.bins_regressor[4])
        if self.norm == "linear":
            # relu + 0.1, normalisation, widths, cumsum and mid-points in one kernel (nnops.bins_head, raw_linear)
            pred = X.bins_head(energy_maps, self.convert_to_prob[0], y, self.min_val, self.max_val, raw_linear=True)
            return {("disp", 0): pred}
        elif self.norm == "softmax":
            return torch.softmax(y, dim=1), energy_maps
        else:
            y = torch.sigmoid(y)
        y = y / y.sum(dim=1, keepdim=True)
        pred = X.bins_head(energy_maps, self.convert_to_prob[0], y, self.min_val, self.max_val)
        return {("disp", 0): pred}


class Depth_Decoder_QueryTr(_QueryTrBase):
    FEED_FORWARD = 1024


class Lite_Depth_Decoder_QueryTr(_QueryTrBase):
    FEED_FORWARD = 512
