from .validation import array2d, check_iter_of_sequences  # noqa: F401
