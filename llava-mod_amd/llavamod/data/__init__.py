from .collate import DataCollatorForDPODataset, DataCollatorForSupervisedDataset  # noqa: F401
