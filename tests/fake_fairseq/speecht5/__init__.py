"""Stand-in for the REFERENCE plug-in package (only its task module's data-plane method), for the load_dataset delegation test."""
