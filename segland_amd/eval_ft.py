#!/usr/bin/env python3
"""eval_ft.py counterpart: evaluation of fine-tuned (base + novel) models, one checkpoint per seed -- see eval_base.py."""
from .eval_base import main as _main


def main(argv=None):
    return _main(argv, ft=True)


if __name__ == '__main__':
    main()
