"""Plugin registry (reference: code/common/model_builder.py): maps `Encoder.Name` and its flags to a
chain of components.  Only the hot-path family is built: `Name=gcn_basis` with UseInputTransform=Yes,
UseOutputTransform=No and none of the experimental layer flags (exactly settings/gcn_block.exp and
settings/gcn_basis.exp); everything else raises NotImplementedError naming SURVEY.md section 2's
out-of-scope row instead of silently building something different."""
from ..decoders.bilinear_diag import BilinearDiag
from ..encoders.affine_transform import AffineTransform
from ..encoders.message_gcns.gcn_basis import BasisGcn
from ..encoders.message_gcns.gcn_basis_concat import ConcatGcn
from ..encoders.relation_embedding import RelationEmbedding
from ..extras.graph_representations import Representation


def _flag(settings, key, default="No"):
    return settings[key] if key in settings else default


def build_encoder(encoder_settings, triples):
    name = encoder_settings['Name']
    if name != "gcn_basis":
        raise NotImplementedError("encoder '%s' is outside the accelerated path (SURVEY.md section 2); "
                                  "only 'gcn_basis' (ConcatGcn / BasisGcn stacks) is built" % name)
    if _flag(encoder_settings, 'UseInputTransform') != "Yes":
        raise NotImplementedError("UseInputTransform=No (one-hot first layer / RandomInput variants)")
    for key in ('UseOutputTransform', 'AddDiagonal', 'DiagonalCoefficients', 'StoreEdgeData', 'RandomInput',
                'PartiallyRandomInput'):
        if _flag(encoder_settings, key) == "Yes":
            raise NotImplementedError("%s=Yes selects a reference variant outside the hot path" % key)
    if _flag(encoder_settings, 'SkipConnections', 'None') != 'None':
        raise NotImplementedError("SkipConnections other than None")

    graph = Representation(triples, encoder_settings)
    input_shape = [int(encoder_settings['EntityCount']), int(encoder_settings['InternalEncoderDimension'])]
    internal_shape = [int(encoder_settings['InternalEncoderDimension']),
                      int(encoder_settings['InternalEncoderDimension'])]
    relation_shape = [int(encoder_settings['EntityCount']), int(encoder_settings['CodeDimension'])]
    if int(encoder_settings['CodeDimension']) != internal_shape[1]:
        raise NotImplementedError("CodeDimension != InternalEncoderDimension needs UseOutputTransform=Yes")
    layers = int(encoder_settings['NumberOfLayers'])

    encoding = AffineTransform(input_shape, encoder_settings, next_component=graph, onehot_input=True,
                               use_bias=True, use_nonlinearity=True)
    encoding = apply_basis_gcn(encoder_settings, encoding, internal_shape, layers)
    return RelationEmbedding(relation_shape, encoder_settings, next_component=encoding)


def apply_basis_gcn(encoder_settings, encoding, internal_shape, layers):
    concat = 'Concatenation' in encoder_settings and encoder_settings['Concatenation'] == "Yes"
    layer_class = ConcatGcn if concat else BasisGcn
    for layer in range(layers):
        encoding = layer_class(internal_shape, encoder_settings, next_component=encoding, onehot_input=False,
                               use_nonlinearity=layer < layers - 1)
    return encoding


def build_decoder(encoder, decoder_settings):
    if decoder_settings['Name'] == "bilinear-diag":
        return BilinearDiag(encoder, decoder_settings)
    raise NotImplementedError("decoder '%s' is not in any BASELINE config (SURVEY.md section 2)"
                              % decoder_settings['Name'])
