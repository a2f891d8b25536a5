from .regression.linear_model import OLS  # noqa: F401
