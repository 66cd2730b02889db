"""Entry point: `python train.py <args_file>` — same contract as the reference's train.py:1-23
(the args file is read through argparse's @file mechanism with whitespace-separated tokens)."""
import sys

from options import MonodepthOptions
from trainer import Trainer


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    options = MonodepthOptions()
    options.parser.convert_arg_line_to_args = lambda line: line.split()
    if len(argv) == 1 and not argv[0].startswith("-"):
        argv = ["@" + argv[0]]
    opts = options.parser.parse_args(argv)
    Trainer(opts).train()


if __name__ == "__main__":
    main()
