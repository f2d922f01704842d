"""Placeholder for import-order side effects shared by the network modules (none needed today)."""
