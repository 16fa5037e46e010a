# -*- coding: utf-8 -*-
"""Error convention of the drop-in boundary: the same exception names the
reference raises (torchkge/exceptions.py:8-45), so user code catching them
keeps working."""


class _KgeError(Exception):
    def __init__(self, message):
        super().__init__(message)


class NotYetEvaluatedError(_KgeError):
    pass


class SizeMismatchError(_KgeError):
    pass


class WrongDimensionError(_KgeError):
    pass


class NotYetImplementedError(_KgeError):
    pass


class WrongArgumentsError(_KgeError):
    pass


class SanityError(_KgeError):
    pass


class SplitabilityError(_KgeError):
    pass


class NoPreTrainedVersionError(_KgeError):
    pass


class NotProvidedError(AttributeError):
    """A name of the reference that this engine leaves out (outside SURVEY.md section 8's hot path).  Raised by the
    packages' module-level ``__getattr__``.  It IS an AttributeError, so ``hasattr(torchkge_amd, name)`` is False and
    ``getattr(torchkge_amd, name, default)`` gives the default, as for any absent attribute; attribute access
    (``torchkge_amd.RelationInference``) shows the message that says what to use instead, and ``from torchkge_amd import
    name`` fails with Python's own ImportError."""
