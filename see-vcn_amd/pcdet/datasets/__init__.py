from .collate import collate_batch, load_data_to_gpu, model_fn_decorator  # noqa: F401
