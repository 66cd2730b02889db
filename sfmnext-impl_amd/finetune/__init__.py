"""Supervised metric-depth finetune path of the reference (finetune/) on the MI355X kernels."""
