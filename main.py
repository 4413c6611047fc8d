"""`python main.py ...` exactly as in the reference README; forwards to linkteller_amd.main."""
from linkteller_amd.main import main

if __name__ == "__main__":
    main()
