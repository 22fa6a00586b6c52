from .reducer import GradReducer  # noqa: F401
