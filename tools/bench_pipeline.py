"""Device-side input pipeline: time one config-B batch (12 samples x 3 frames, 375x1242 -> 192x640, flip + colour jitter + ToTensor)
on the device, and the same frames through PIL on one host core for reference.  usage: python tools/bench_pipeline.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sqd.pipeline import DevicePreprocess, draw_params  # noqa: E402

B, F, H0, W0, H, W = 12, 3, 375, 1242, 192, 640
rs = np.random.RandomState(0)
raw = torch.from_numpy(rs.randint(0, 256, (B, F, H0, W0, 3)).astype(np.uint8)).cuda()
rng = np.random.default_rng(1)
params = [draw_params(rng) for _ in range(B)]
params[0] = (True, ([0, 1, 2, 3], 1.1, 0.9, 1.05, -0.03))
pre = DevicePreprocess(H, W)
for _ in range(3):
    out = pre(raw, [p[0] for p in params], [p[1] for p in params])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 20
for _ in range(N):
    out = pre(raw, [p[0] for p in params], [p[1] for p in params])
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print("device: %.3f ms per batch of %d frames (%.0f frames/s); bytes in %.1f MB" % (ms, B * F, B * F / ms * 1e3, raw.numel() / 1e6))
try:
    from PIL import Image, ImageEnhance
    cpu = raw.cpu().numpy()
    t0 = time.perf_counter()
    for b in range(B):
        for f in range(F):
            im = Image.fromarray(cpu[b, f])
            if params[b][0]:
                im = im.transpose(Image.FLIP_LEFT_RIGHT)
            im = im.resize((W, H), Image.LANCZOS)
            a = params[b][1]
            if a is not None:
                im2 = ImageEnhance.Brightness(im).enhance(a[1])
                im2 = ImageEnhance.Contrast(im2).enhance(a[2])
                im2 = ImageEnhance.Color(im2).enhance(a[3])
                im2 = im2.convert("HSV").convert("RGB")
            _ = np.asarray(im, dtype=np.float32) / 255.0
    dt = time.perf_counter() - t0
    print("PIL on one host core (resize + enhancers + HSV round trip, no hue shift arithmetic): %.1f ms per batch (%.0f frames/s)" % (dt * 1e3, B * F / dt))
except ImportError:
    pass
