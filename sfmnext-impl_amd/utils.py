"""Small host helpers the trainer surface uses (reference utils.py: readlines, normalize_image,
sec_to_hm_str)."""


def readlines(filename):
    with open(filename, "r") as f:
        return f.read().splitlines()


def normalize_image(x):
    """Rescale an image tensor to [0,1] for visualisation."""
    ma, mi = float(x.max().cpu().data), float(x.min().cpu().data)
    d = ma - mi if ma != mi else 1e5
    return (x - mi) / d


def sec_to_hm(t):
    t = int(t)
    s = t % 60
    t //= 60
    return t // 60, t % 60, s


def sec_to_hm_str(t):
    h, m, s = sec_to_hm(t)
    return "{:02d}h{:02d}m{:02d}s".format(h, m, s)
