from . import ops, util  # noqa: F401
