"""CATER / Something-Else vocabularies.

Restates the dictionaries built in the reference dataset constructors
(data/cater.py:95-124, data/smth.py:85-104) so that the model surface can be
constructed without the datasets (no network / no video files here).
"""


def _finish(vocab):
    vocab['pred_idx_to_name'] = {v: k for k, v in vocab['pred_name_to_idx'].items()}
    vocab['action_idx_to_name'] = {v: k for k, v in vocab['action_name_to_idx'].items()}
    vocab['reverse_attributes'] = {a: {v: k for k, v in d.items()} for a, d in vocab['attributes'].items()}
    # object_name_to_idx: "__image__" -> 0, every other attribute label gets "<label>_<running index>"
    # (data/cater.py:114-124)
    o2i, ind = {}, 0
    for attr in vocab['attributes']:
        for label in vocab['attributes'][attr]:
            o2i[label if ind == 0 else "{}_{}".format(label, ind)] = ind
            ind += 1
    vocab['object_name_to_idx'] = o2i
    return vocab


def cater_vocab():
    v = {}
    v['pred_name_to_idx'] = {'__in_image__': 0, 'right': 1, 'above': 2, 'below': 3, 'left': 4,
                             'surrounding': 5, 'inside': 6, '__padding__': 7}
    v['action_name_to_idx'] = {'__in_image__': 0, '_no_op': 1, '_slide': 2, '_contain': 3, '_rotate': 4,
                               '_pick_place': 5, '__padding__': 6}
    v['attributes'] = {
        'shape': {'__image__': 0, 'cube': 1, 'sphere': 2, 'cylinder': 3, 'spl': 4, 'cone': 5},
        'color': {'__image__': 0, 'gray': 1, 'red': 2, 'blue': 3, 'green': 4, 'brown': 5,
                  'purple': 6, 'cyan': 7, 'yellow': 8, 'gold': 9},
        'material': {'__image__': 0, 'rubber': 1, 'metal': 2},
        'size': {'__image__': 0, 'small': 1, 'large': 2, 'medium': 3},
    }
    return _finish(v)


def smth_vocab(num_objects=30):
    """Something-Else shaped vocabulary (1 attribute => semantic_nc = 128), data/smth.py:85-104."""
    v = {}
    v['pred_name_to_idx'] = {'__in_image__': 0, 'right': 1, 'above': 2, 'below': 3, 'left': 4,
                             'surrounding': 5, 'inside': 6, 'cover': 7, '__padding__': 8}
    names = ['__in_image__', 'putting', 'pushing', 'pulling', 'moving', 'taking', 'lifting', 'covering',
             '__padding__']
    v['action_name_to_idx'] = {n: i for i, n in enumerate(names)}
    objs = {'__image__': 0}
    for i in range(1, num_objects):
        objs['obj_%d' % i] = i
    objs['hand'] = num_objects
    v['attributes'] = {'object': objs}
    return _finish(v)
