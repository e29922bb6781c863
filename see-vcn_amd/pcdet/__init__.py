"""Mirror of the reference's `pcdet` package for the hot path (same registries, names, batch_dict contract)."""
