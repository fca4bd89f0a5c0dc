"""Error types of the reference's configuration layer (exceptions/exceptions.py:1-10)."""


class ConfigurationError(Exception):
    """A configuration value is missing, inconsistent or unknown (parser/parser.py:39-40,152)."""


class InitializationError(Exception):
    """An object was used before it was initialised (parser/parser.py:102-103)."""
