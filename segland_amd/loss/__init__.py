from .criterion import OrthLoss


def get_loss(args):
    """loss/__init__.py:3-8 of the reference: POP models train with OrthLoss."""
    if 'pop' in args.model:
        return OrthLoss(ignore_index=args.ignore_label)
    raise RuntimeError('segland_amd implements the POP path only (model %r): CELoss models are out of scope' % (args.model,))
