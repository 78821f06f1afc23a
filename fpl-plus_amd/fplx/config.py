"""INI config parser - same dialect and typing rules as the reference's
PyMIC/pymic/util/parse_config.py:7-111 (sections -> dict[section][lower-case key] -> int / float /
list / bool / None / str by literal shape), so the reference's .cfg files parse unchanged."""
import configparser


def is_int(val_str):
    start = 1 if val_str[0] == '-' else 0
    return all('0' <= ch <= '9' for ch in val_str[start:])


def is_float(val_str):
    if '.' in val_str and len(val_str.split('.')) == 2 and './' not in val_str:
        a, b = val_str.split('.')
        return is_int(a) and is_int(b)
    if 'e' in val_str and val_str[0] != 'e' and len(val_str.split('e')) == 2:
        a, b = val_str.split('e')
        return is_int(a) and is_int(b)
    return False


def is_bool(var_str):
    return var_str.lower() in ('true', 'false')


def parse_bool(var_str):
    return var_str.lower() == 'true'


def is_list(val_str):
    return val_str[0] == '[' and val_str[-1] == ']'


def _scalar(item, in_list):
    if is_int(item):
        return int(item)
    if is_float(item):
        return float(item)
    if not in_list and is_list(item):
        return parse_list(item)
    if is_bool(item):
        return parse_bool(item)
    if item.lower() == 'none':
        return None
    return item


def parse_list(val_str):
    return [_scalar(item.strip(), True) for item in val_str[1:-1].split(',')]


def parse_value_from_string(val_str):
    return _scalar(val_str, False)


def parse_config(filename):
    config = configparser.ConfigParser()
    config.read(filename)
    output = {}
    for section in config.sections():
        output[section] = {}
        for key in config[section]:
            val_str = str(config[section][key])
            if len(val_str) > 0:
                output[section][key] = parse_value_from_string(val_str)
    return output


def synchronize_config(config):
    """parse_config.py:102-111"""
    data_cfg, net_cfg = config['dataset'], config['network']
    data_cfg["LabelToProbability_class_num".lower()] = net_cfg["class_num"]
    if "PartialLabelToProbability" in data_cfg.get('train_transform', []):
        data_cfg["PartialLabelToProbability_class_num".lower()] = net_cfg["class_num"]
    return config
