"""Reference-compatible flag module (`from Parameters import parse_args`); see mimrl_amd/Parameters.py."""
from mimrl_amd.Parameters import build_parser, parse_args  # noqa: F401

if __name__ == "__main__":
    print(parse_args())
