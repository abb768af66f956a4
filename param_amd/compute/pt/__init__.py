"""Mirror of the reference's train/compute/pt embedding driver (CLI + stdout rows)."""
