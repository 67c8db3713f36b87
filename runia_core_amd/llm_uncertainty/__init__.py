from .scores import eigen_score, semantic_entropy  # noqa: F401

__all__ = ["eigen_score", "semantic_entropy"]
