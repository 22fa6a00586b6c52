from .reducer import GradReducer, partition_modules  # noqa: F401
