kcal = 4.184 * 1000.0 / 96485.33212 / 1.0 * 1.0  # placeholder; not used on the hot path
mol = 6.02214076e23
