"""associaTR: TR length x phenotype association scan (mirror of trtools/associaTR)."""
