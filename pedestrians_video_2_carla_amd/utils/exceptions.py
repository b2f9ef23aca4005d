class NotAvailableException(Exception):
    """An optional (third-party) model is not installed (reference utils/exceptions.py:1-10)."""

    def __init__(self, model_name: str, flow_name: str = None):
        where = f' in flow {flow_name}' if flow_name else ''
        super().__init__(f'{model_name} is not available{where}: its third-party implementation is not installed.')
