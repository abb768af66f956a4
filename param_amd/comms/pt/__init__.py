"""Mirror of the reference's train/comms/pt plug-in surface for the DLRM all-to-all path."""
