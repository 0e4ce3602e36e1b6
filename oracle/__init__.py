"""CPU restatement of the reference env step -- TEST INFRASTRUCTURE ONLY (see dcc_oracle.c)."""
