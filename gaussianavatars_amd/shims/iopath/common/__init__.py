from . import file_io  # noqa: F401
