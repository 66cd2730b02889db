"""Self Query Layer module (reference networks/layers.py:4-21): parameter-free, so the module is a thin
shell over the operator in sqd.nnops."""
import torch.nn as nn

from sqd import nnops as X


class FullQueryLayer(nn.Module):
    def forward(self, x, K):
        """x [bs,E,H,W], K [bs,Q,E] -> (energy maps [bs,Q,H,W], summary embeddings [bs,Q,E])."""
        assert x.shape[1] == K.shape[2], \
            "Number of channels in x and Embedding dimension (at dim 2) of K matrix must match"
        return X.full_query_layer(x, K)
