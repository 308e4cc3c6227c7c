from .import_utils import is_xformers_available  # noqa: F401
