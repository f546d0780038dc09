"""CPU oracle of the hot path (test infrastructure only): qd_oracle.c / oracle.py (env step), policy_ref.py (policy networks)."""
