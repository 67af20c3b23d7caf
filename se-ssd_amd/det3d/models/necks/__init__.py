from .rpn_v1 import RPN, SSFA  # noqa: F401
