"""`python Main.py <reference flags>` -- same command line as the reference's Main.py; see mimrl_amd/Main.py."""
from mimrl_amd.Main import main

if __name__ == "__main__":
    main()
