from .scores import eigen_score  # noqa: F401

__all__ = ["eigen_score"]
