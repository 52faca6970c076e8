from .phiseg import PHISeg  # noqa: F401
