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
