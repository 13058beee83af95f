"""nutpie_amd — MI355X-native NUTS engine behind nutpie's Python API."""
from nutpie_amd._lib import __version__  # noqa: F401
