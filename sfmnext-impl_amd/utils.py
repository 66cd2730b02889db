"""Small host helpers the trainer surface uses (the names of the reference's utils.py: readlines, normalize_image, sec_to_hm,
sec_to_hm_str — split files, tensorboard images, the log line's time-left field)."""
import datetime


def readlines(filename):
    """the lines of a text file (a split file: one sample per line), without their terminators"""
    with open(filename) as fh:
        return [line.rstrip("\r\n") for line in fh]


def normalize_image(x):
    """an image tensor stretched to [0, 1] over its own value range (tensorboard); a constant image maps to 0, as in the reference"""
    lo, hi = x.min(), x.max()
    span = float(hi - lo)
    return (x - lo) / (span if span != 0.0 else 1e5)


def sec_to_hm(t):
    """seconds -> (hours, minutes, seconds), whole numbers"""
    minutes, seconds = divmod(int(t), 60)
    hours, minutes = divmod(minutes, 60)
    return hours, minutes, seconds


def sec_to_hm_str(t):
    """seconds -> '10h06m02s'-style text"""
    return "%02dh%02dm%02ds" % sec_to_hm(t)
